"""Build-time guard for the kernels that issue loads from inline asm with hand-counted waits (csrc/conv1d.hip gemm_rows_kernel and conv_splitk_f32_kernel):
the compiler treats an asm output as defined when the asm statement ends, so under register pressure it may copy or spill the
destination registers before the data has landed (seen in this repository: a `v_accvgpr_write` right behind an asm `ds_read`).
The test compiles the source to gfx950 assembly and checks, for every asm load, that no compiler-generated instruction reads its
destination registers before the next asm `s_waitcnt vmcnt` of the same basic block (the copy-right-behind-the-asm failure), and that
the kernels carry no scratch spills."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_reads_an_asm_load_destination_before_its_wait(tmp_path):
    src = os.path.join(ROOT, "ims-toucan-prosody-variance_amd", "csrc", "conv1d.hip")
    out = tmp_path / "conv1d.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), src], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    names = re.findall(r"^(_ZN3tts16gemm_rows_kernel\S+|_ZN3tts22conv_splitk_f32_kernel\S+):", text, re.M)
    assert len(names) >= 6  # four gemm_rows instantiations + the two split-K ones
    for name in names:
        body = text[text.index("\n" + name + ":"):]
        body = body[: body.index(".Lfunc_end")]
        assert "scratch_" not in body, f"{name}: spills"
        lines = body.split("\n")
        in_asm, pending, loads = False, [], 0  # pending: destination register sets of asm loads not yet waited for
        for ln in lines:
            code = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
            if code.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if code.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if code.endswith(":") or code.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                pending = []  # basic-block local: behind a label or a branch the counted waits of the other paths apply
                continue
            if not code or code.startswith("."):
                continue
            if in_asm:
                if code.startswith("global_load"):
                    pending.append(_regs(code.split(",")[0]))
                    loads += 1
                elif code.startswith("s_waitcnt") and "vmcnt" in code:
                    pending = []  # (conservative: the first counted wait behind a load ends its window)
                continue
            if not pending:
                continue
            ops = code.split(None, 1)
            if len(ops) < 2:
                continue
            fields = [f.strip() for f in ops[1].split(",")]
            stores = ops[0].startswith(("global_store", "ds_write", "scratch_store", "buffer_store"))
            sources = fields if stores else fields[1:]
            read = set().union(*[_regs(f) for f in sources]) if sources else set()
            for dest in pending:
                assert not (read & dest), f"{name}: `{code}` reads v{sorted(read & dest)} while an asm load into them is in flight"
        assert loads > 0, name
