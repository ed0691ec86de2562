"""Build libpulse_hip.so (gfx950) in-tree with hipcc.

    python pulse_amd/csrc/build.py [--force] [--verbose]

hipcc cross-compiles gfx950 without a GPU.  Objects and the shared library are
written next to the sources (git-ignored, but they travel to the GPU box with
the gpurun snapshot).  Each translation unit is rebuilt only when it or a header
is newer than its object.
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libpulse_hip.so")
ARCH = "gfx950"

# (source, extra flags).  The parity-critical elementwise kernels are built without FMA
# contraction so sums/products round like the eager PyTorch reference.
NO_CONTRACT = ["-ffp-contract=off"]
SOURCES = [
    ("capi.cpp", []),
    ("rot_ops.hip", NO_CONTRACT),
    ("env_step.hip", NO_CONTRACT),
    ("amp_obs.hip", NO_CONTRACT),
    ("task_step.hip", NO_CONTRACT),
    ("motion_state.hip", NO_CONTRACT),
    ("rollout_ops.hip", NO_CONTRACT),
    ("gae.hip", NO_CONTRACT),
    # MFMA accumulators in VGPR form: no v_accvgpr moves (VALU slots are what the fp32 MFMA loop is short of) and the
    # whole kernel fits the 256-register budget of two waves per SIMD
    ("gemm_f32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]),   # packed-f32 VALU (v_pk_add_f32) beside MFMAs costs more than two plain adds
    ("learner_ops.hip", NO_CONTRACT),
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    cc = hipcc()
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "pulse_hip.h"))
    headers.append(os.path.abspath(__file__))
    common = ["-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
              "-Wno-unused-function", f"-I{INCLUDE}", f"-I{HERE}"]
    jobs = []
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(HERE, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(HERE, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            jobs.append((src, [cc] + common + extra + ["-c", s, "-o", o]))

    def run(job):
        name, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return name, r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for name, rc, out in ex.map(run, jobs):
            if out.strip() and (verbose or rc != 0):
                print(f"--- {name} ---\n{out}", flush=True)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {name}")
    if jobs or force or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
