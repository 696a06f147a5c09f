#!/usr/bin/env python3
"""CLI for symbols-from-video_amd/isa_check.py: check_tr_asm.py file.s [kernel-prefix read-op ...]"""
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location(
    "isa_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "symbols-from-video_amd", "isa_check.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
args = sys.argv[2:]
bad = mod.tr_asm_hazards(open(sys.argv[1]).read(), *( [args[0], tuple(args[1:])] if args else [] ))
print("\n".join(bad[:40]))
print("asm-read hazards:", len(bad))
sys.exit(1 if bad else 0)
