"""CPU: the C-ABI library loads and exports every symbol include/rbvae_hip.h declares."""
import ctypes
import os

import sfv_amd


def test_library_exports_every_declared_symbol():
    L = sfv_amd._lib
    assert os.path.exists(L.LIB_PATH), "build librbvae_hip.so first (__graft_entry__.build())"
    protos = L.parse_header()
    assert len(protos) >= 10
    raw = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in protos if not hasattr(raw, n)]
    assert not missing, missing
    assert L.query("rbvae_version") >= 100
    assert isinstance(L.lib().rbvae_last_error(), bytes)
    # the product library carries no debug probes; they live in librbvae_dbg.so behind include/rbvae_dbg.h
    assert not [n for n in dir(raw) if n.startswith("rbvae_dbg_")] and not hasattr(raw, "rbvae_dbg_mfma_bf16")
    dbg = ctypes.CDLL(L.DBG_LIB_PATH)
    probes = [n for n in L.parse_header(L.DBG_HEADER) if "stamps" not in n]
    assert len(probes) == 6 and all(hasattr(dbg, n) for n in probes)
