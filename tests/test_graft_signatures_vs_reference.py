"""CPU, build container only: every method `install()` grafts onto the reference's classes must be callable the way the reference's
own method of that name is -- same positional parameter names in the same order, a default wherever the reference has one, and
nothing but defaulted parameters added behind.  The reference sources are PARSED (ast; nothing is imported, executed, copied or
shipped); the test skips where /root/reference does not exist (the GPU box)."""
import ast
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fisher-nerf-customized_amd")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

# (reference file, reference class, product file, product mixin class(es) that hold the grafted methods)
TARGETS = [
    ("models/SLAM/gaussian.py", "GaussianSLAM", "models/SLAM/gaussian.py", ("FisherOps",)),
    ("models/SLAM/gaussian_object.py", "GaussianObjectSLAM", "models/SLAM/gaussian_object.py", ("ObjectFisherOps",)),
    ("planning/astar.py", "AstarPlanner", "planning/astar.py", ("OccupancyOps",)),
]
# what the product adds that the reference class does not have (helpers and new operators): allowed, but named here so that a
# typo in a grafted name cannot hide as "new"
NEW_IN_PRODUCT = {
    "FisherOps": {"_device", "_as_w2c", "_stack_poses", "_scorer", "_scorer_key", "_keyframe_key", "_same_keyframes", "_PARAM_KEYS", "path_scores"},
    "ObjectFisherOps": {"_draw_probes", "_probe_rows", "_pose_probe_rows", "_flat_diag", "_diag_batch", "_diag_scores", "_block_columns",
                        "_visible_indices", "_block_scores"},
    "OccupancyOps": {"_fbe_point", "_ring_candidates", "generate_candidate_in_freespace", "_next_seed", "_eroded_free", "filter_candidates_in_freespace", "cells_of", "_occ_cfg",
                     "_occ_workspace", "_stream"},
}


def _classes(path):
    tree = ast.parse(open(path).read(), filename=path)
    return {n.name: n for n in ast.walk(tree) if isinstance(n, ast.ClassDef)}


def _methods(cls_node):
    return {n.name: n for n in cls_node.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))}


def _signature(fn):
    """(positional names without self, number of them that have defaults, keyword-only names, has *args, has **kwargs)"""
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    if pos and pos[0] in ("self", "cls"):
        pos = pos[1:]
    return pos, len(a.defaults), [x.arg for x in a.kwonlyargs], a.vararg is not None, a.kwarg is not None


def _grafted_names(install_fn):
    """the string constants of the tuples `install` iterates over (for name in (...): setattr(...))"""
    names = []
    for node in ast.walk(install_fn):
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Tuple):
            names += [e.value for e in node.iter.elts if isinstance(e, ast.Constant) and isinstance(e.value, str)]
    return names


@pytest.mark.parametrize("ref_file,ref_cls,our_file,mixins", TARGETS)
def test_grafted_methods_accept_the_reference_call_forms(ref_file, ref_cls, our_file, mixins):
    ref_classes = _classes(os.path.join(REF, ref_file))
    assert ref_cls in ref_classes, f"{ref_cls} not found in {ref_file}"
    ref_methods = _methods(ref_classes[ref_cls])
    ours = _classes(os.path.join(PKG, our_file))
    checked = 0
    for mixin in mixins:
        ours_methods = _methods(ours[mixin])
        assert "install" in ours_methods, mixin
        grafted = _grafted_names(ours_methods["install"])
        assert grafted, f"{mixin}.install grafts nothing?"
        for name in grafted:
            if name not in ref_methods:
                assert name in NEW_IN_PRODUCT[mixin], f"{mixin}.install grafts `{name}`, which {ref_cls} does not define and the test does not list as new"
                continue
            assert name in ours_methods, f"{mixin} has no method `{name}` to graft"
            r_pos, r_ndef, r_kwonly, r_var, r_kw = _signature(ref_methods[name])
            o_pos, o_ndef, o_kwonly, o_var, o_kw = _signature(ours_methods[name])
            # positional call forms: the reference's parameters, by name and order, lead ours
            assert o_pos[:len(r_pos)] == r_pos, f"{ref_cls}.{name}: reference parameters {r_pos}, graft {o_pos}"
            # a parameter the reference lets the caller omit must be omittable here
            r_required = len(r_pos) - r_ndef
            o_required = len(o_pos) - o_ndef
            assert o_required <= r_required, f"{ref_cls}.{name}: graft requires {o_pos[:o_required]}, reference only {r_pos[:r_required]}"
            # keyword-only parameters of the reference (rare) must exist; ours added behind must all be optional (checked by o_required)
            for k in r_kwonly:
                assert k in o_kwonly or k in o_pos or o_kw, f"{ref_cls}.{name}: keyword `{k}` of the reference is not accepted"
            if r_var:
                assert o_var, f"{ref_cls}.{name}: the reference takes *args"
            checked += 1
    assert checked >= 3, f"only {checked} grafted methods of {ref_cls} were compared"


def test_the_reference_call_sites_pass_only_arguments_the_grafts_take():
    """The call sites the grafts serve (tester_gaussians_navigation.py:1423-2204): every keyword a call of compute_Hessian /
    compute_H_train / pose_eval* passes there is a parameter of the grafted method, and no call passes more positional arguments
    than it takes."""
    src = open(os.path.join(REF, "tester_gaussians_navigation.py")).read()
    tree = ast.parse(src)
    ours = {}
    ours.update(_methods(_classes(os.path.join(PKG, "models/SLAM/gaussian.py"))["FisherOps"]))
    obj = _methods(_classes(os.path.join(PKG, "models/SLAM/gaussian_object.py"))["ObjectFisherOps"])
    used = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)}
    assert {"compute_Hessian", "compute_H_train", "pose_eval"} <= used
    n_calls = 0
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)):
            continue
        name = node.func.attr
        if name not in ("compute_Hessian", "compute_H_train", "pose_eval", "pose_eval_popgs", "compute_H_train_popgs"):
            continue
        for table in (ours, obj):
            if name not in table:
                continue
            pos, _, kwonly, var, kw = _signature(table[name])
            assert var or len(node.args) <= len(pos), f"line {node.lineno}: {name} called with {len(node.args)} positional arguments, graft takes {pos}"
            for k in node.keywords:
                assert k.arg is None or kw or k.arg in pos or k.arg in kwonly, f"line {node.lineno}: {name}({k.arg}=...) is not a parameter of the graft {pos}"
            n_calls += 1
    assert n_calls >= 8
