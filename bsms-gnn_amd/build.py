"""Build libbsms_hip.so for gfx950 with hipcc, in-tree (the .so travels to the GPU box with the
snapshot).  Usage: python bsms-gnn_amd/build.py [--force]"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libbsms_hip.so")
SOURCES = ["plan.hip", "rowsum.hip", "chain.hip", "efuse.hip", "efwd.hip", "wgrad.hip", "gmp.hip", "bsgmp.hip", "optim.hip", "hierarchy.hip", "sim.hip"]
# csrc/experiments/*.hip (the fp32 fused edge backward, efuse32.hip) are NOT part of the product library: profiles/build_efv.sh
# compiles them into the experiment builds (-DBSMS_EXPERIMENTS), where gmp.hip / chain.hip keep their hooks
HEADERS = ["common.h", "chain.h", "chain_dev.h", os.path.join("..", "..", "include", "bsms_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("BSMS_EXPERIMENTS") == "1":   # profiling / A-B builds only: BSMS_DEBUG_FLAGS, in-kernel time stamps, launch-shape knobs
    FLAGS.append("-DBSMS_EXPERIMENTS")
    SOURCES = SOURCES + [os.path.join("experiments", "efuse32.hip")]   # the hooks gmp.hip keeps under BSMS_EXPERIMENTS need it to link


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256((" ".join(FLAGS) + os.environ.get("BSMS_CHAIN_FLAGS", "")).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
        # no contraction: rowsum.hip rounds x*ew before the add like the reference; sim.hip keeps the fp64 normaliser roundings
        extra = ["-ffp-contract=off"] if src in ("rowsum.hip", "sim.hip") else []
        if src == "chain.hip" and os.environ.get("BSMS_CHAIN_FLAGS"):   # A/B builds
            extra += os.environ["BSMS_CHAIN_FLAGS"].split()
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(digest)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


ASAN_DIR = os.path.join(OBJ, "asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libbsms_host_asan.so")
ASAN_SOURCES = ["plan.hip", "hierarchy.hip"]     # the library's HOST code: plan pool / recycling / uploads, hierarchy builder


def asan_runtime():
    """Path of GCC's shared AddressSanitizer runtime (LD_PRELOADed into the torch-free driver process), or None.
    GCC's, not the one shipped with ROCm's clang: that runtime intercepts hsa_amd_memory_pool_allocate for DEVICE-side
    ASan (needs xnack) and makes every hipMalloc fail with 'out of memory' on this platform."""
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = os.path.realpath(r.stdout.strip()) if r.returncode == 0 else ""
    return path if path and os.path.isabs(path) and os.path.exists(path) else None


def build_host_sanitized(force=False):
    """ASan + UBSan build of the library's host-only translation units (plan.hip, hierarchy.hip contain no device code)
    with g++ as plain C++ against the HIP runtime API, as a separate small library that only the sanitizer tests load
    (tests/helpers/host_sanitizer_driver.py).  The plan pool, the retirement of plans and the builder run on several host
    threads: lifetime bugs there corrupt device index blocks silently, which is what a sanitizer run is for (SURVEY.md
    section 5: race detection / sanitizers)."""
    os.makedirs(ASAN_DIR, exist_ok=True)
    stamp = os.path.join(ASAN_DIR, "stamp")
    h = hashlib.sha256(b"asan-gcc-v2")
    for f in ASAN_SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    digest = h.hexdigest()
    if not force and os.path.exists(ASAN_LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return ASAN_LIB
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    objs = []
    for src in ASAN_SOURCES:
        obj = os.path.join(ASAN_DIR, src.replace(".hip", ".o"))
        r = subprocess.run(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *san,
                            "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ (sanitized) failed on {src}:\n{r.stdout}\n{r.stderr}")
        objs.append(obj)
    r = subprocess.run(["g++", "-shared", "-fPIC", *san, "-o", ASAN_LIB, *objs, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link (sanitized) failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(digest)
    return ASAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--asan" in sys.argv:
        print(build_host_sanitized(force="--force" in sys.argv))
