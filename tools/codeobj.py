"""tools/codeobj.py -- identity of ONE kernel's machine code inside a built libfisher_rast.so (measurement aid, no GPU needed).

The hardware-counter record that bench.py quotes (profiles/pmc_k_fisher_tile_v3.json) is only valid for the kernel binary it
was collected on.  It is stamped with `kernel_code_id(so, "k_fisher_tile_v3")` = sha256 of that kernel's gfx950 instructions,
read out of the shared object itself (the embedded clang offload bundle -> the gfx950 code object's ELF symbol table -> the
function's bytes); bench.py recomputes the id from the library it has loaded and uses the record only when the two agree.  A
change anywhere else in the sources leaves the id alone, a change to the kernel (or to the compiler flags) cannot.

    python tools/codeobj.py [path/to/libfisher_rast.so] [kernel-name-substring ...]
"""
import hashlib
import os
import struct
import sys

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob, arch="gfx950"):
    """the device ELF images of every offload bundle in the file whose target triple names `arch`"""
    out = []
    pos = blob.find(_MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        q = pos + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode("ascii", "replace")
            q += 24 + tl
            if arch in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(_MAGIC, pos + 1)
    return out


def _functions(elf):
    """{symbol name: machine code bytes} of the FUNC symbols of one ELF64 little-endian image"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1, "not an ELF64 LE image"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]   # name type flags addr off size link info align entsize
    funcs = {}
    for s in sec:
        if s[1] != 2:          # SHT_SYMTAB
            continue
        strtab = sec[s[6]]
        for k in range(s[5] // 24):
            name_off, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s[4] + 24 * k)
            if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= shnum:      # STT_FUNC, defined
                continue
            end = elf.index(b"\0", strtab[4] + name_off)
            name = elf[strtab[4] + name_off:end].decode("ascii", "replace")
            sh = sec[shndx]
            start = sh[4] + (value - sh[3])
            funcs[name] = elf[start:start + size]
    return funcs


def kernel_code_id(so_path, kernel_substring, arch="gfx950"):
    """sha256 (16 hex digits) over the machine code of every kernel of the library whose mangled name contains `kernel_substring`
    (template instantiations sorted by name), or None when there is none."""
    blob = open(so_path, "rb").read()
    found = {}
    for co in _code_objects(blob, arch):
        for name, code in _functions(co).items():
            if kernel_substring in name:
                found[name] = code
    if not found:
        return None
    h = hashlib.sha256()
    for name in sorted(found):
        h.update(name.encode()); h.update(struct.pack("<Q", len(found[name]))); h.update(found[name])
    return h.hexdigest()[:16]


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith(".so") else os.path.join(root, "fisher-nerf-customized_amd", "fisher_rast", "libfisher_rast.so")
    for k in (args or ["k_fisher_tile_v4", "k_preprocess_views", "k_sort_tiles"]):
        print(k, kernel_code_id(so, k))
