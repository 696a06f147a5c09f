"""The build-time ISA check (symbols-from-video_amd/isa_check.py) on hand-written instruction streams:
it must accept the guarded pattern the GEMM kernels use and reject every way a fragment register can be
touched before the wait that covers its asynchronous LDS read."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("isa_check", os.path.join(HERE, "..", "symbols-from-video_amd", "isa_check.py"))
isa_check = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_check)

PFX = "_ZN5rbvae12wgrad_gemm_k"


def kernel(body, asm_ops=("ds_read_b64_tr_b16",)):
    """the fragment reads as the kernels issue them: inline asm (between the compiler's ;;#ASMSTART / ;;#ASMEND marks)"""
    lines = []
    for ln in body.splitlines():
        if ln.strip().split(" ")[0] in asm_ops:
            lines += ["\t;;#ASMSTART", ln, "\t;;#ASMEND"]
        else:
            lines.append(ln)
    return f"{PFX}ItLi2ELi3EEEvNS_6WgArgsE:\n" + "\n".join(lines) + "\n\ts_endpgm\n"


def check(body):
    return isa_check.tr_asm_hazards(kernel(body), PFX, ("ds_read_b64_tr_b16",))


def test_guarded_pipeline_is_clean():
    body = """
	ds_read_b64_tr_b16 v[0:1], v40
	ds_read_b64_tr_b16 v[2:3], v40 offset:1024
.LBB0_1:
	ds_read_b64_tr_b16 v[4:5], v41
	ds_read_b64_tr_b16 v[6:7], v41 offset:1024
	s_waitcnt lgkmcnt(2)
	v_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[0:3], v[20:23]
	s_waitcnt lgkmcnt(0)
	s_cbranch_scc1 .LBB0_3
	s_barrier
	ds_read_b64_tr_b16 v[0:1], v40
	ds_read_b64_tr_b16 v[2:3], v40 offset:1024
.LBB0_3:
	v_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[4:7], v[20:23]
	s_cbranch_scc0 .LBB0_1
	s_waitcnt lgkmcnt(0)
	global_store_dwordx4 v[30:31], v[20:23], off
"""
    assert check(body) == []


def test_copy_before_wait_is_flagged():
    body = """
	ds_read_b64_tr_b16 v[0:1], v40
	v_mov_b64_e32 v[8:9], v[0:1]
	s_waitcnt lgkmcnt(0)
"""
    bad = check(body)
    assert len(bad) == 1 and "touches in-flight" in bad[0]


def test_counted_wait_releases_only_the_oldest():
    body = """
	ds_read_b64_tr_b16 v[0:1], v40
	ds_read_b64_tr_b16 v[2:3], v40 offset:1024
	s_waitcnt lgkmcnt(1)
	v_mov_b32_e32 v9, v0
	v_mov_b32_e32 v10, v2
	s_waitcnt lgkmcnt(0)
"""
    bad = check(body)
    assert len(bad) == 1 and "v[2]" in bad[0]


def test_hazard_through_the_loop_back_edge():
    # the read at the bottom of the loop is still in flight when the top of the next iteration uses v0
    body = """
.LBB0_1:
	v_add_u32_e32 v9, v0, v9
	ds_read_b64_tr_b16 v[0:1], v40
	s_cbranch_scc0 .LBB0_1
	s_waitcnt lgkmcnt(0)
"""
    bad = check(body)
    assert any("touches in-flight v[0]" in b for b in bad)


def test_in_flight_register_as_address_or_second_destination():
    body = """
	ds_read_b64_tr_b16 v[0:1], v40
	ds_read_b64_tr_b16 v[2:3], v1
	ds_read_b64_tr_b16 v[0:1], v41
	s_waitcnt lgkmcnt(0)
"""
    bad = check(body)
    assert any("used as address" in b for b in bad) and any("second read" in b for b in bad)


def test_built_kernels_are_clean_when_the_listing_exists():
    # build() writes the listings next to the objects; absent on a checkout that has not been built
    for src, (prefix, ops) in (("wgrad_gemm", (PFX, ("ds_read_b64_tr_b16",))),
                               ("gather_gemm", ("_ZN5rbvae13gather_gemm_k", ("ds_read_b128",)))):
        path = os.path.join(HERE, "..", "symbols-from-video_amd", "build", src + ".s")
        if os.path.exists(path):
            assert isa_check.tr_asm_hazards(open(path).read(), prefix, ops) == []


def test_asm_vmem_load_check_flags_a_copy_before_the_wait():
    """isa_check.asm_vmem_load_hazards (conv_halo_k's register-staged patch pieces): a compiler move of an asm load's
    destination before the counted vmcnt wait is flagged; LDS-DMA and stores count in the queue but carry no registers;
    the same code with the move behind the wait is clean."""
    import importlib
    ic = importlib.import_module("symbols-from-video_amd.isa_check")
    head = "_ZN5rbvae11conv_halo_kTEST:\n"
    body_bad = (head +
                "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
                "\tglobal_load_lds_dwordx4 v[4:5], off\n"
                "\tv_mov_b32_e32 v20, v11\n"
                "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n"
                "\ts_endpgm\n")
    bad = ic.asm_vmem_load_hazards(body_bad, "_ZN5rbvae11conv_halo_k")
    assert len(bad) == 1 and "v_mov_b32_e32 v20, v11" in bad[0]
    body_ok = body_bad.replace("\tv_mov_b32_e32 v20, v11\n", "") .replace("\ts_endpgm", "\tv_mov_b32_e32 v20, v11\n\ts_endpgm")
    assert ic.asm_vmem_load_hazards(body_ok, "_ZN5rbvae11conv_halo_k") == []
    # the wait must really cover the load: two younger operations, vmcnt(2) leaves it in flight
    body_short = body_ok.replace("vmcnt(1)", "vmcnt(2)")
    assert ic.asm_vmem_load_hazards(body_short, "_ZN5rbvae11conv_halo_k")
    # a loop-carried destination copied on the back edge before the wait (the bug this check was written for)
    loop = (head + ".LBB0_1:\n"
            "\tv_mov_b32_e32 v30, v10\n"
            "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"
            "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
            "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert any("v_mov_b32_e32 v30, v10" in m for m in ic.asm_vmem_load_hazards(loop, "_ZN5rbvae11conv_halo_k"))


def test_compiler_reads_take_a_place_in_the_queue_but_carry_no_registers():
    """a compiler-issued LDS read (outside the asm marks) is the compiler's to wait for: its counted wait behind a mix of
    its own reads is read for what it covers, and its destination is never 'in flight' for this check"""
    body = """
	ds_read_b128 v[14:17], v19 offset:16384
	ds_read_b32 v21, v1
	ds_read_b128 v[10:13], v19 offset:16400
	s_waitcnt lgkmcnt(2)
	v_add_u32_e32 v18, v15, v14
"""
    assert isa_check.tr_asm_hazards(kernel(body, asm_ops=()), PFX, ("ds_read_b128",)) == []
    # the same first read as inline asm, waited for one operation short: flagged
    bad = isa_check.tr_asm_hazards(kernel(body.replace("lgkmcnt(2)", "lgkmcnt(3)"), asm_ops=("ds_read_b128",)), PFX, ("ds_read_b128",))
    assert any("touches in-flight" in b for b in bad)
    # ... and with the exact count it is clean (the ds_read_b32 between them holds its place in the in-order queue)
    assert isa_check.tr_asm_hazards(kernel(body, asm_ops=("ds_read_b128",)), PFX, ("ds_read_b128",)) == []
