"""Build-time check of the wgrad kernels' ISA.

wgrad_gemm_k issues its transposed LDS reads (ds_read_b64_tr_b16) as inline asm so that the compiler's
wait-count pass does not drain the LDS-DMA ring in front of them; the price is that the compiler no longer
knows those registers are filled asynchronously.  The source guards every group with a tied s_waitcnt; this
check proves it on the generated code: between such a read and the wait that covers it no instruction may
touch a register the read is still filling (the hardware does not interlock them)."""
import re


def tr_asm_hazards(asm_text, kernel_prefix="_ZN5rbvae12wgrad_gemm_k"):
    """Returns a list of 'kernel: message' strings (empty = clean).  Linear scan in text order, which is
    conservative for this kernel: every loop exit passes a full s_waitcnt lgkmcnt(0)."""
    out = []
    for m in re.finditer(r"^(" + re.escape(kernel_prefix) + r"\w+):.*?s_endpgm", asm_text, re.S | re.M):
        reads = []          # in-flight transposed reads, oldest first (each: set of destination registers)
        for ln in m.group(0).splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or ln.endswith(":") or ln.startswith("."):
                continue
            op, _, rest = ln.partition(" ")
            if op == "ds_read_b64_tr_b16":
                d = re.match(r"\s*v\[(\d+):(\d+)\]", rest)
                regs = set(range(int(d.group(1)), int(d.group(2)) + 1))
                uses = set(int(x) for x in re.findall(r"\bv(\d+)\b", rest.split(",", 1)[1]))
                pending = set().union(*reads) if reads else set()
                if uses & pending:
                    out.append(f"{m.group(1)}: in-flight register used as address: {ln}")
                reads.append(regs)
                continue
            if op == "s_waitcnt" and "lgkmcnt" in rest:
                n = int(re.search(r"lgkmcnt\((\d+)\)", rest).group(1))
                # LDS returns in order: at most the n youngest operations are still outstanding (other LGKM
                # operations in between only make this more conservative)
                reads = reads[len(reads) - n:] if n else []
                continue
            if not reads:
                continue
            pending = set().union(*reads)
            used = set()
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", rest):
                used |= set(range(int(a), int(b) + 1))
            used |= set(int(x) for x in re.findall(r"\bv(\d+)\b", rest))
            if used & pending:
                out.append(f"{m.group(1)}: touches in-flight v{sorted(used & pending)}: {ln}")
    return out
