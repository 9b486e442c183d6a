"""Drop-in for the `simple_knn` package the reference imports (`from simple_knn._C import distCUDA2`,
gs3dgs/scene/gaussian_model.py:22)."""
