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
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libpulse_hip.so")
ARCH = "gfx950"

# (source, extra flags).  The parity-critical elementwise kernels are built without FMA
# contraction so sums/products round like the eager PyTorch reference.
#
# EVERY translation unit is built with -fno-slp-vectorize, and the build fails if a packed-fp32 VALU instruction (v_pk_mul_f32 /
# v_pk_add_f32 / v_pk_fma_f32) shows up in the device code anyway.  Round 3 finding (tools/im_step_repro.py):
# on gfx950 / ROCm 7.2 those instructions return WRONG values in the last quarter of the wave (lanes 48-63) while a wave of another
# kernel issues MFMAs on the same SIMD -- the fused env step (632 of them after SLP vectorisation) gave observation rows that
# differed by O(1) from launch to launch on identical inputs whenever a second process ran GEMMs on the GPU, and was bit-stable
# alone, beside a VALU / copy competitor, at -O1, or with SLP vectorisation off.  Scalar fp32 VALU code is not affected.
NO_CONTRACT = ["-ffp-contract=off"]
NO_PACKED_F32 = ["-fno-slp-vectorize", "-fno-vectorize"]      # SLP and loop vectoriser both form <2 x float> arithmetic
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
SOURCES = [
    ("capi.cpp", []),
    ("rot_ops.hip", NO_CONTRACT),
    ("env_step.hip", NO_CONTRACT),
    ("amp_obs.hip", NO_CONTRACT),
    ("task_step.hip", NO_CONTRACT),
    ("traj_step.hip", NO_CONTRACT),
    ("motion_state.hip", NO_CONTRACT),
    ("rollout_ops.hip", NO_CONTRACT),
    ("gae.hip", NO_CONTRACT),
    # MFMA accumulators in VGPR form: no v_accvgpr moves (VALU slots are what the fp32 MFMA loop is short of) and the
    # whole kernel fits the 256-register budget of two waves per SIMD
    ("gemm_f32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),   # (packed-f32 VALU beside MFMAs also costs more than two plain adds)
    ("gemm_x3p.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
    # the 256 x 256 x3 tile runs ONE wave per SIMD with its 256 accumulator registers in AGPRs: no VGPR-form flag here
    # (its k-tile body is one fully unrolled 96-MFMA schedule: lift the pragma-unroll size limit, or the loop stays rolled and the accumulators go to scratch)
    ("gemm_x3w.hip", ["-mllvm", "-pragma-unroll-threshold=4000000", "-Wno-unused-const-variable"]),
    # the skinny-N x3 kernel: one wave per SIMD with plenty of registers (a phase of raw A + B staging units in flight)
    ("gemm_x3s.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
    ("learner_ops.hip", NO_CONTRACT),
    ("b16_ops.hip", NO_CONTRACT),
    ("vae_head.hip", NO_CONTRACT),
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


HOST_ONLY_OBJECTS = ("capi.o",)          # the only translation unit without device code: everything else must yield a gfx950 code object


def llvm_bin():
    """The LLVM tools that belong to the hipcc in use (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), with the stock location as fallback."""
    cc = hipcc()
    cands = []
    if os.path.isabs(cc):
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(cc))), "lib", "llvm", "bin"))
    cands.append(LLVM_BIN)
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    raise RuntimeError("llvm-objdump / llvm-objcopy / clang-offload-bundler not found next to hipcc: the device-ISA audit cannot run")


def packed_f32_instructions(obj, host_only=False):
    """Disassemble the gfx950 code object embedded in a host object and return its packed-fp32 arithmetic instructions (must be none)."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        llvm = llvm_bin()
        r = subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            if host_only:
                return []                                       # capi.cpp: no device code
            # the contention-safety argument rests on this audit: an object whose device code cannot be looked at FAILS
            raise RuntimeError(f"{obj}: no .hip_fatbin section could be extracted ({r.stderr.strip() or 'empty section'}); cannot audit its device ISA")
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}",
                        f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            raise RuntimeError(f"{obj}: the offload bundle holds no {ARCH} code object; cannot audit its device ISA")
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    return re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b[^\n]*", dis)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    cc = hipcc()
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "pulse_hip.h"))
    headers.append(os.path.abspath(__file__))
    common = ["-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
              "-Wno-unused-function", f"-I{INCLUDE}", f"-I{HERE}"] + NO_PACKED_F32
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
    for o in objs:
        bad = packed_f32_instructions(o, host_only=os.path.basename(o) in HOST_ONLY_OBJECTS)
        if bad:
            raise RuntimeError(f"{os.path.basename(o)}: {len(bad)} packed-fp32 VALU instructions in the device code (e.g. {bad[0].strip()}); "
                               "they are unsafe beside MFMA waves on gfx950 (see the note at the top of build.py)")
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
