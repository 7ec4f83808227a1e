"""Drop-in for the `simple_knn` package (see _C.distCUDA2)."""
