"""Build libsslam_frontend.so (HIP, gfx950 only) in-tree with hipcc.

    python structure-slam-pointline_amd/build.py [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot."""
import os, subprocess, sys, glob

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsslam_frontend.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("SSLAM_EXTRA_FLAGS", "").split()


# per-unit flags: the NFA stage without machine LICM (csrc/lines_nfa.hip says why)
UNIT_FLAGS = {"lines_nfa.hip": ["-mllvm", "-disable-machine-licm"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(HERE, "..", "include", "sslam_frontend.h")]
    return any(os.path.getmtime(d) > t for d in deps)


# libsslam_frontend_testing.so = the same sources with -DSSLAM_TESTING: the product's entry points plus the self-tests, probes and the RCCL stand-in declared in
# include/sslam_testing.h.  Only these units hold SSLAM_TESTING code and are compiled a second time; the other objects are shared.
TEST_LIB = os.path.join(LIBDIR, "libsslam_frontend_testing.so")
TESTING_UNITS = ("lines.hip", "group.hip")


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not needs_build() and os.path.exists(TEST_LIB) and os.path.getmtime(TEST_LIB) >= os.path.getmtime(LIB):
        return LIB
    objs, tobjs = [], []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in sources():
        base = os.path.basename(s)
        o = os.path.join(objdir, base + ".o")
        objs.append(o)
        jobs.append([HIPCC] + FLAGS + UNIT_FLAGS.get(base, []) + ["-c", s, "-o", o])
        if base in TESTING_UNITS:
            t = os.path.join(objdir, base + ".testing.o")
            tobjs.append(t)
            jobs.append([HIPCC] + FLAGS + UNIT_FLAGS.get(base, []) + ["-DSSLAM_TESTING", "-c", s, "-o", t])
        else:
            tobjs.append(o)
    ncpu = max(1, min(len(jobs), (os.cpu_count() or 4)))
    running = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd), flush=True)
        running.append((cmd, subprocess.Popen(cmd)))
        while len([1 for _, p in running if p.poll() is None]) >= ncpu:
            for _, p in running:
                if p.poll() is None:
                    p.wait(); break
    for cmd, p in running:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    for lib, oo in ((LIB, objs), (TEST_LIB, tobjs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + oo
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


SHIM = os.path.join(HERE, "shim")
SHIM_LIB = os.path.join(LIBDIR, "libsslam_shim.so")
SHIM_TEST = os.path.join(LIBDIR, "shim_test")


def build_shim(force=False, verbose=True):
    """The C++ host side (reference class surfaces over the C ABI): plain g++, links the HIP library."""
    build(force=False, verbose=verbose)
    srcs = [os.path.join(SHIM, f) for f in ("shim.cc", "shim_test.cpp", "cv_min.h", "ORBextractor.h", "ExtractLineSegment.h", "FrontendMatchers.h", "ORBmatcher.h", "LSDmatcher.h")]
    if not force and os.path.exists(SHIM_TEST) and os.path.exists(SHIM_LIB) and \
            all(os.path.getmtime(x) < os.path.getmtime(SHIM_TEST) for x in srcs + [LIB]):
        return SHIM_TEST
    cxx = os.environ.get("CXX", "g++")
    cmds = [[cxx, "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared", os.path.join(SHIM, "shim.cc"), "-L" + LIBDIR, "-lsslam_frontend",
             "-Wl,-rpath,$ORIGIN", "-o", SHIM_LIB],
            [cxx, "-O2", "-std=c++17", "-Wall", os.path.join(SHIM, "shim_test.cpp"), "-L" + LIBDIR, "-lsslam_shim", "-lsslam_frontend",
             "-Wl,-rpath,$ORIGIN", "-o", SHIM_TEST]]
    for c in cmds:
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
    return SHIM_TEST


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_shim(force="--force" in sys.argv))
