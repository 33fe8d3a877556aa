"""Builds the compiled extension modules voxelocc / gputransform / voxelfeat over libmrslam_hip.so.

    python bindings/cython/setup.py build_ext --build-lib bindings/cython/_built

Replaces the three reference build scripts that locate nvcc and compile kernel.cu + manager.cu
(generate_bev_cython_binary/setup.py:20-111 and its two siblings): here the extension only links the C-ABI library."""
import os

import numpy as np
from Cython.Build import cythonize
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIBDIR = os.path.join(ROOT, "mr_slam_amd")

ext = [Extension(name, [os.path.join(HERE, name + ".pyx")],
                 include_dirs=[os.path.join(ROOT, "include"), np.get_include(), HERE],
                 libraries=["mrslam_hip"], library_dirs=[LIBDIR], runtime_library_dirs=[LIBDIR],
                 define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")])
       for name in ("voxelocc", "gputransform", "voxelfeat")]

setup(name="mrslam_bev_modules", ext_modules=cythonize(ext, include_path=[HERE], build_dir=os.path.join(HERE, "_cython_c"),
                                                      compiler_directives={"language_level": 3}), script_args=None)
