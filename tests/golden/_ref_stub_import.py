"""
In-container only: import the *reference* tIGAr.BSplines from /root/reference with
stub modules standing in for dolfin / petsc4py / ufl (absent in this image), so that
golden input/output vectors can be generated from the reference's own arithmetic
(SURVEY.md section 8c).  Nothing here travels to the GPU box except the fixtures
(*.npz) this produces; the reference source is never copied into the repo.

The reference's inline C++ routine ``basisFuncsInner`` (tIGAr/BSplines.py:48-127) is
compiled from the string where it lies, with the one unused dolfin include dropped
and the pybind11 module name substituted -- that is the only arithmetic native code
of the reference and it runs unmodified.
"""
import os, sys, types, tempfile, subprocess, importlib.util, sysconfig

REF_ROOT = "/root/reference"


def _compile_cpp_code(code, **kw):
    import pybind11
    name = "tigar_ref_basisfuncs"
    d = tempfile.mkdtemp(prefix="tigar_ref_")
    src = os.path.join(d, name + ".cpp")
    code = code.replace("#include <dolfin/common/Array.h>", "")
    code = code.replace("SIGNATURE", name)
    with open(src, "w") as f:
        f.write(code)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(d, name + ext)
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           src, "-o", out]
    subprocess.check_call(cmd)
    spec = importlib.util.spec_from_file_location(name, out)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    sys.dont_write_bytecode = True
    dolfin = types.ModuleType("dolfin")

    class _Params(dict):
        pass
    dolfin.parameters = _Params(linear_algebra_backend="PETSc")

    class _Comm(object):
        pass

    class _MPI(object):
        comm_world = _Comm()
        comm_self = _Comm()

        @staticmethod
        def size(c):
            return 1

        @staticmethod
        def rank(c):
            return 0
    dolfin.MPI = _MPI
    dolfin.DOLFIN_EPS = 3.0e-16

    def near(a, b, eps=3.0e-16):
        return abs(a - b) <= eps   # dolfin::near: |x - x0| <= eps  [ext]
    dolfin.near = near
    dolfin.compile_cpp_code = _compile_cpp_code

    class NonlinearProblem(object):
        pass

    class NewtonSolver(object):
        pass

    class SubDomain(object):
        pass
    dolfin.NonlinearProblem = NonlinearProblem
    dolfin.NewtonSolver = NewtonSolver
    dolfin.SubDomain = SubDomain
    sys.modules["dolfin"] = dolfin

    petsc4py = types.ModuleType("petsc4py")
    petsc4py.init = lambda *a, **k: None
    PETSc = types.ModuleType("petsc4py.PETSc")
    petsc4py.PETSc = PETSc
    sys.modules["petsc4py"] = petsc4py
    sys.modules["petsc4py.PETSc"] = PETSc

    ufl = types.ModuleType("ufl")
    ufl.indices = ufl.rank = ufl.shape = None
    ufl_eq = types.ModuleType("ufl.equation")

    class Equation(object):
        pass
    ufl_eq.Equation = Equation
    ufl.equation = ufl_eq
    ufl_cl = types.ModuleType("ufl.classes")
    ufl_cl.PermutationSymbol = None
    ufl.classes = ufl_cl
    sys.modules["ufl"] = ufl
    sys.modules["ufl.equation"] = ufl_eq
    sys.modules["ufl.classes"] = ufl_cl


def import_reference():
    """Returns the reference module tIGAr.BSplines (with tIGAr.common loaded)."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import tIGAr.BSplines as B
    return B
