"""Build ``micronet_amd/lib/libmicronet_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU).

    python -m micronet_amd.build [--force]

Flags: ``-ffp-contract=off`` (the integer quantize step must round exactly like the reference's separate ATen ops),
IEEE-correct fp32 division is hipcc's default (never -ffast-math).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libmicronet_hip.so")
SOURCES = ["quant_kernels.hip", "conv_kernels.hip", "qgemm_kernels.hip", "qgemm_kxk.hip", "qgemm_sign.hip", "qgemm_pwb.hip", "qgemm_k3s.hip", "qgemm_dense.hip", "conv_first.hip", "optim_kernels.hip", "norm_kernels.hip", "iao_ops.hip", "qact_kernels.hip", "data_kernels.hip", "linear_kernels.hip", "iao_bnfuse.hip", "iao_g3.hip", "iao_thin.hip"]
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 operations of DIFFERENT staged rows into v_pk_* instructions and pays for it with register shuffles on the loop
# back edge, each behind a wait for the prefetched loads -- the software pipelines of k_pws_wgrad_s (round 3) and k_pwb (round 6: vmcnt(2) instead of vmcnt(27) in
# every step) drained every iteration.  Round 6 A/B of the whole library with and without it: c3 55.6k -> 58.3k img/s (k_pw_wgrad<4,4,2,2> 531 -> 439 us per step,
# k_bf_gram<4,0> 402 -> 287), c2 +0.9 %, c1_w2a2 / c4 / c5 unchanged: every file is built without it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize"]
EXTRA_FLAGS = {}


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "micronet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # one hipcc process per source file, a few at a time (the container has 8 cores; a file takes 10-60 s)
    from concurrent.futures import ThreadPoolExecutor
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src.replace(".hip", ".o"))
        cmds.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    jobs = max(1, min(int(os.environ.get("MN_BUILD_JOBS", "6")), os.cpu_count() or 1))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(run, cmds))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
