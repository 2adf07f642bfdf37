"""Drop-in `simple_knn` (thirdparty/simple-knn of the reference, an un-vendored submodule)."""
