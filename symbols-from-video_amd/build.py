"""Build librbvae_hip.so (the product C-ABI, include/rbvae_hip.h) and librbvae_dbg.so (hardware-map probes,
include/rbvae_dbg.h) for gfx950 with hipcc, in-tree, next to this file.
RBVAE_DEBUG=1: also build librbvae_hip_debug.so with the GEMM kernels' phase stamps compiled in (-DGG_STAMPS=1
-DWG_STAMPS=1; select it with RBVAE_LIB=...)."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librbvae_hip.so")
DBG_LIB = os.path.join(HERE, "librbvae_dbg.so")
DEBUG_LIB = os.path.join(HERE, "librbvae_hip_debug.so")
DBG_SOURCES = ("dbg.hip",)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, extra=(), suffix=""):
    obj = os.path.join(OBJ, src[:-4] + suffix + ".o")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "rbvae_hip.h"))
    if _stale(obj, [os.path.join(CSRC, src)] + headers):
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if src in ISA_CHECKED and not extra:
            try:
                _check_asm_reads(src)
            except Exception:
                os.remove(obj)          # a failed check must not leave an object the next build would link
                raise
        return obj, True
    return obj, False


# kernels whose LDS fragment reads are inline asm: (mangled-name prefix, read opcodes)
ISA_CHECKED = {
    "wgrad_gemm.hip": ("_ZN5rbvae12wgrad_gemm_k", ("ds_read_b64_tr_b16",)),
    "gather_gemm.hip": ("_ZN5rbvae13gather_gemm_k", ("ds_read_b128",)),
    "conv_halo.hip": ("_ZN5rbvae11conv_halo_k", ("ds_read_b128",)),
    "conv_halo_ws.hip": ("_ZN5rbvae14conv_halo_ws_k", ("ds_read_b128",)),
    "deconv_halo.hip": ("_ZN5rbvae13deconv_halo_k", ("ds_read_b128",)),
    "wgrad_halo.hip": ("_ZN5rbvae12wgrad_halo_k", ("ds_read_b64_tr_b16",)),
    "wgrad_row.hip": ("_ZN5rbvae11wgrad_row_k", ("ds_read_b64_tr_b16",)),
    "conv_s2.hip": ("_ZN5rbvae9conv_s2_k", ("ds_read_b128",)),
    "conv_first.hip": (("_ZN5rbvae13wgrad_first_k", "_ZN5rbvae18wgrad_first_wide_k"), ("ds_read_b64_tr_b16",)),
}


# kernels with register-destination loads issued as inline asm (counted vmcnt waits): the load opcode
ASM_VMEM_LOADS = {"conv_halo.hip": "global_load_dwordx4", "deconv_halo.hip": "global_load_dwordx4",
                  "conv_halo_ws.hip": "buffer_load_dwordx4"}


def _check_asm_reads(src):
    """The GEMM kernels hide their LDS fragment reads from the compiler (inline asm); prove on the ISA that no
    fragment register is touched before the wait that covers it (isa_check.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbvae_isa_check", os.path.join(HERE, "isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = os.path.join(OBJ, src[:-4] + ".s")
    cmd = [HIPCC] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stderr}")
    prefixes, ops = ISA_CHECKED[src]
    bad = []
    for prefix in ((prefixes,) if isinstance(prefixes, str) else prefixes):
        bad += mod.tr_asm_hazards(open(asm).read(), prefix, ops)
        if src in ASM_VMEM_LOADS:
            bad += mod.asm_vmem_load_hazards(open(asm).read(), prefix, ASM_VMEM_LOADS[src])
    if bad:
        raise RuntimeError(src + " ISA check failed (fragment register touched before its wait):\n" + "\n".join(bad[:20]))


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        res = list(ex.map(_compile, sources()))
    objs = [o for (o, _), src in zip(res, sources()) if src not in DBG_SOURCES]
    dbg_objs = [o for (o, _), src in zip(res, sources()) if src in DBG_SOURCES]

    def link(target, objects, extra=()):
        if _stale(target, objects) or any(c for _, c in res):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objects + list(extra)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
            if verbose:
                print("built", target)

    link(LIB, objs)
    # the probes report errors through the main library's rbvae::fail
    link(DBG_LIB, dbg_objs, ["-L" + HERE, "-lrbvae_hip", "-Wl,-rpath,$ORIGIN"])
    if os.environ.get("RBVAE_DEBUG") == "1":
        stamped = ["-DGG_STAMPS=1", "-DWG_STAMPS=1"]
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
            dres = list(ex.map(lambda s_: _compile(s_, stamped, ".dbg"), [s_ for s_ in sources() if s_ not in DBG_SOURCES]))
        link(DEBUG_LIB, [o for o, _ in dres])
    return LIB


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
