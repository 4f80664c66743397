"""Build libgiga_hip.so in-tree with hipcc for gfx950:  python -m giga_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libgiga_hip.so")
DEMO_SRC = os.path.join(os.path.dirname(HERE), "examples", "c_abi_demo.cpp")
DEMO = os.path.join(HERE, "lib", "c_abi_demo")
PEAK_SRC = os.path.join(os.path.dirname(HERE), "tools", "mfma_peak.hip")
PEAK = os.path.join(HERE, "lib", "mfma_peak")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", "Makefile"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "giga_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    if force or _stale():
        r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=not verbose, text=True)
        if r.returncode != 0:
            sys.stderr.write((r.stdout or "") + (r.stderr or ""))
            raise RuntimeError("hipcc build of libgiga_hip.so failed")
    if not os.path.exists(LIB):
        raise RuntimeError("libgiga_hip.so missing after build")
    # the torch-free C++ host example (examples/c_abi_demo.cpp) links against the library it demonstrates
    if os.path.exists(DEMO_SRC) and (force or not os.path.exists(DEMO) or
                                     os.path.getmtime(DEMO) < max(os.path.getmtime(DEMO_SRC), os.path.getmtime(LIB))):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        r = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", DEMO_SRC, "-o", DEMO,
                            "-L" + os.path.dirname(LIB), "-lgiga_hip", "-Wl,-rpath,$ORIGIN"], capture_output=not verbose, text=True)
        if r.returncode != 0:
            sys.stderr.write((r.stdout or "") + (r.stderr or ""))
            raise RuntimeError("hipcc build of examples/c_abi_demo.cpp failed")
    # the standalone micro-benchmarks behind the hardware figures quoted in DESIGN.md (tools/*.hip): sustained MFMA / HBM
    # peaks, fp32 MFMA vs VALU co-execution and per-instruction issue cost, the XCD-local barrier, f16 MFMA denormals, the issue cost
    # of the f16 chains' VALU instructions, the read rate of a layer-to-layer hand-off through one XCD's L2
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tools = os.path.join(os.path.dirname(HERE), "tools")
    for name in ("mfma_peak", "mfma_valu_overlap", "mfma_issue_cost", "xcd_barrier", "mfma_denorm", "valu_f16_cost", "l2_handoff", "l2_atomic_rate", "tr16_probe"):
        src, out = os.path.join(tools, name + ".hip"), os.path.join(HERE, "lib", name)
        if os.path.exists(src) and (force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src)):
            r = subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", src, "-o", out], capture_output=not verbose, text=True)
            if r.returncode != 0:
                sys.stderr.write((r.stdout or "") + (r.stderr or ""))
                raise RuntimeError(f"hipcc build of tools/{name}.hip failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
