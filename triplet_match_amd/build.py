"""Build recipe for libtm_hip.so (hipcc, gfx950 only, in-tree).

Flags that are part of the numerics contract (DESIGN.md "Numerics"):
  -ffp-contract=off                           no FMA contraction on host or device
  -fhip-fp32-correctly-rounded-divide-sqrt    IEEE division / sqrt on the device
  (no -ffast-math, denormals preserved)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtm_hip.so")
SOURCES = ["tm_core.hip", "tm_model.hip", "tm_scene.hip", "tm_scene_build.hip", "tm_traits.hip", "tm_earlydrop.hip", "tm_comm.hip", "tm_cloud.hip"]
HEADERS = ["tm_math.h", "tm_internal.h", "tm_table.h", "tm_traits_math.h", "tm_score.h", "tm_icp_math.h", "tm_knn.h",
           os.path.join("..", "..", "include", "tm_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


EXPERIMENTS_LIB = os.path.join(HERE, "libtm_hip_experiments.so")


def build(force=False, verbose=False, timing_experiments=False):
    """Compile every HIP translation unit for gfx950 and link libtm_hip.so. Returns its path.

    timing_experiments=True builds a SEPARATE library, libtm_hip_experiments.so, with -DTM_TIMING_EXPERIMENTS:
    the stage-ablation switches (option "ablate": wrong results, timing only) exist only there; select it with
    TM_LIB=<path> (capi.py). The product library never contains them."""
    if timing_experiments:
        return _build(EXPERIMENTS_LIB, ["-DTM_TIMING_EXPERIMENTS"], ".exp.o", verbose)
    if not force and not _stale():
        return LIB
    return _build(LIB, [], ".o", verbose)


def build_variant(name, defines, verbose=False):
    """A/B aid: libtm_hip_<name>.so built with extra -D switches (select it with TM_LIB=<path>); never the product."""
    return _build(os.path.join(HERE, "libtm_hip_%s.so" % name), list(defines), ".%s.o" % name, verbose)


def _build(lib, extra, obj_suffix, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", obj_suffix))
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-pthread"]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m triplet_match_amd.build --variant NAME -DTM_X=1 ...
        k = sys.argv.index("--variant")
        print(build_variant(sys.argv[k + 1], [a for a in sys.argv[k + 2:] if a.startswith("-D")], verbose=False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, timing_experiments="--timing-experiments" in sys.argv))
