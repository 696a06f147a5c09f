"""ctypes binding of librbvae_hip.so, generated from include/rbvae_hip.h.

The product path has NO fallback: if the shared library is missing or a symbol
the header declares is absent, importing a kernel fails loudly."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "rbvae_hip.h")
LIB_PATH = os.environ.get("RBVAE_LIB") or os.path.join(HERE, "librbvae_hip.so")   # RBVAE_LIB: A/B another build

E_INVALID, E_LAUNCH, E_UNSUPPORTED = -1, -2, -3

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "unsigned long long": ctypes.c_ulonglong, "unsigned": ctypes.c_uint, "double": ctypes.c_double,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname)])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(rbvae_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    parsed.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    parsed.append((_CTYPES[ty.replace("const ", "")], nm))
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        protos[name] = (restype, parsed)
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). There is no CPU or PyTorch fallback for the RBVAE hot path.")
        l = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (restype, args) in _protos.items():
            fn = getattr(l, name)           # AttributeError = header/library drift: fail loudly
            fn.restype = restype
            fn.argtypes = [t for t, _ in args]
        _lib = l
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return x


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Call an int-returning entry point on torch's current stream; raise on error."""
    l = lib()
    fn = getattr(l, name)
    rc = fn(*[_ptr(a) for a in args], stream_ptr())
    if rc != 0:
        msg = l.rbvae_last_error().decode()
        if rc == E_INVALID:
            raise ValueError(f"{name}: {msg}")
        raise RuntimeError(f"{name} failed ({rc}): {msg}")


def query(name, *args):
    return getattr(lib(), name)(*args)


# ---- diagnostic probes (include/rbvae_dbg.h, librbvae_dbg.so): not part of the product ABI ----------------
DBG_HEADER = os.path.join(os.path.dirname(HERE), "include", "rbvae_dbg.h")
DBG_LIB_PATH = os.path.join(HERE, "librbvae_dbg.so")
_dbg = None


def dbg_lib():
    """librbvae_dbg.so itself (for the hooks that take no stream, e.g. rbvae_dbg_conv_halo_variant)."""
    global _dbg
    if _dbg is None:
        lib()                                   # the probes resolve rbvae::fail (and the hooks their switches) from the main library
        l = ctypes.CDLL(DBG_LIB_PATH)
        for n, (restype, a) in parse_header(DBG_HEADER).items():
            fn = getattr(l, n, None)            # the stamp hooks exist in stamped builds of the main library only
            if fn is not None:
                fn.restype = restype
                fn.argtypes = [t for t, _ in a]
        _dbg = l
    return _dbg


def dbg_call(name, *args):
    """Call a hardware-map probe of librbvae_dbg.so on torch's current stream."""
    global _dbg
    if _dbg is None:
        lib()                                   # the probes resolve rbvae::fail from the main library
        l = ctypes.CDLL(DBG_LIB_PATH)
        for n, (restype, a) in parse_header(DBG_HEADER).items():
            fn = getattr(l, n, None)            # the stamp hooks exist in stamped builds of the main library only
            if fn is not None:
                fn.restype = restype
                fn.argtypes = [t for t, _ in a]
        _dbg = l
    fn = getattr(_dbg, name, None)
    if fn is None:                              # a stamp hook: lives in the stamped build of the main library (RBVAE_LIB)
        fn = getattr(lib(), name)
        fn.restype, fn.argtypes = parse_header(DBG_HEADER)[name][0], [t for t, _ in parse_header(DBG_HEADER)[name][1]]
    rc = fn(*[_ptr(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().rbvae_last_error().decode()}")
