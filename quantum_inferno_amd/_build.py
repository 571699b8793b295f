"""Builds libqi_tfr.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

The library is linked against the HIP runtime and hipFFT that PyTorch-ROCm ships
(torch/lib), not the copies under /opt/rocm/lib: device pointers and streams cross the
C ABI from torch, so both sides must share ONE HIP runtime in the process.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libqi_tfr.so")
SOURCES = ["qi_api.hip", "qi_api_ops.hip", "qi_host_util.hip", "qi_plan_build.hip", "qi_run.hip", "qi_kernels.hip", "qi_native.hip", "qi_block.hip", "qi_zoom.hip", "qi_shannon1d.hip", "qi_stft_sliding.hip", "qi_stft_fused.hip", "qi_zoom64.hip"]
ARCH = "gfx950"
# the FFT kernels lose ~10 % to the register shuffles of SLP-packed v_pk_* arithmetic (no throughput gain on gfx950)
PER_FILE_FLAGS = {"qi_native.hip": ("-fno-slp-vectorize",), "qi_block.hip": ("-fno-slp-vectorize",), "qi_zoom.hip": ("-fno-slp-vectorize",)}


def torch_lib_dir():
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        raise RuntimeError("PyTorch-ROCm is required (its bundled HIP runtime is the one the library links)")
    return os.path.join(list(spec.submodule_search_locations)[0], "lib")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".o")]
    deps += [os.path.join(ROOT, "include", "qi_tfr.h"), os.path.abspath(__file__)]  # (the flags live in this file)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=(), lib=None):
    lib = lib or LIB
    if not force and not extra_flags and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tlib = torch_lib_dir()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".dbg.o" if extra_flags else ".o"))
        cmd = [hipcc, "-c", "-fPIC", "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, "-Wall", "-Wno-unused-function", *PER_FILE_FLAGS.get(src, ()), *extra_flags,
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    link = ["g++", "-shared", "-o", lib, *objs, f"-L{tlib}", "-lamdhip64", "-lhipfft", f"-Wl,-rpath,{tlib}",
            "-Wl,--no-undefined", "-lpthread"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    if "--stamps" in sys.argv:  # diagnostic library with in-kernel phase stamps (never the shipped one)
        print(build(force=True, extra_flags=("-DQI_NATIVE_STAMPS", "-DQI_NATIVE_DEBUG"),
                    lib=os.path.join(HERE, "libqi_tfr_stamps.so")))
    elif "--variant" in sys.argv:  # experiment library: --variant NAME -DFLAG ... -> libqi_tfr_NAME.so
        i = sys.argv.index("--variant")
        print(build(force=True, extra_flags=tuple(sys.argv[i + 2:]),
                    lib=os.path.join(HERE, f"libqi_tfr_{sys.argv[i + 1]}.so")))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
