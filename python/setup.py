"""Builds the SWIG module for a real OpenMM install:  OPENMM_DIR=... VV_DIR=<install prefix of this plugin> python setup.py install
(run `swig -python -c++ -I$OPENMM_DIR/include -I../openmmapi/include -o VVPluginWrapper.cpp velocityverletplugin.i` first)."""
import os
from setuptools import Extension, setup

openmm_dir = os.environ.get("OPENMM_DIR", "/usr/local/openmm")
vv_dir = os.environ.get("VV_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
setup(name="velocityverletplugin", version="1.0",
      py_modules=["velocityverletplugin"],
      ext_modules=[Extension(name="_velocityverletplugin", sources=["VVPluginWrapper.cpp"],
                             libraries=["OpenMM", "OpenMMDrude", "OpenMMVelocityVerlet"],
                             include_dirs=[os.path.join(openmm_dir, "include"), os.path.join(vv_dir, "openmmapi", "include")],
                             library_dirs=[os.path.join(openmm_dir, "lib"), os.path.join(vv_dir, "lib")],
                             runtime_library_dirs=[os.path.join(openmm_dir, "lib"), os.path.join(vv_dir, "lib")],
                             extra_compile_args=["-std=c++17"])])
