"""Static checks on the compiler's output for gfx950 (no GPU needed: hipcc cross-compiles here) -- the code-generation accidents that cost
this project real time in round 3 (docs/history/DESIGN_r1-r4.md 3.11, 3.12), as regression tests:
  * no FLAT instruction in any kernel: a pointer that has been through an empty asm statement loses its address space, its accesses become
    flat_load / flat_store, which are slower and complete out of order with the other memory counters;
  * the production march kernels keep scratch out of their per-tile sample loops (and spill at most a handful of per-block values);
  * no instruction touches a matrix-instruction RESULT inside 11 wait states (tools/mfma_war_check.py): the compiler pads that for its own
    instructions but not for inline asm (the in-place relu), docs/history/DESIGN_r1-r4.md 3.5.  (The script's other check -- early writes of an A / B OPERAND --
    is informational since round 3: tools/ubench/mfma_war.hip shows that gfx950 does not read operands after issue.)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = [("hav_ops", []), ("hav_train", []), ("hav_mlp_train", []), ("hav_conv", []), ("hav_render", ["-DHAV_FAST_BUILD"])]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp("isa")
    for name, extra in SOURCES:
        dst = str(d / (name + ".s"))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", dst,
                        os.path.join(ROOT, "havatar_amd", "csrc", name + ".hip")] + extra, check=True, stderr=subprocess.DEVNULL, timeout=600)
        out[name] = dst
    return out


def test_no_flat_instructions_in_any_kernel(asm):
    for name, path in asm.items():
        bad = [l.strip() for l in open(path) if re.match(r"\s+flat_(load|store|atomic)", l)]
        assert not bad, "%s: %d FLAT instructions, e.g. %s" % (name, len(bad), bad[:3])


def _kernel_body(text, sym):
    a = text.index("\n" + sym + ":")
    return text[a:text.index("s_endpgm", a)].splitlines()


def _innermost_matrix_loops(lines):
    """[(first, last)] line ranges of the innermost loops (a backward branch to a label) that contain matrix instructions: the per-tile
    sample loops of the march kernel."""
    label = {}
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", l.strip())
        if m:
            label[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i:
            a, b = label[m.group(1)], i
            if any("v_mfma" in x for x in lines[a:b]):
                loops.append((a, b))
    return [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]


def test_production_march_kernels_keep_scratch_out_of_the_tile_loops_and_pass_the_static_hazard_check(asm):
    """The production kernels -- <0|1, 3, 2> (fp16 x 2 + MX: the default arithmetic), <0|1, 1, 2> (bf16 triple split) and <0|1, 2, 2> (fp16
    double split): NO spilled VGPR and no scratch at all (pinned: a compiler change that brings spills back fails here), and their per-tile
    sample loops (the innermost loops that hold matrix instructions: gather, layers, heads, compositing) hold the expected matrix work."""
    text = open(asm["hav_render"]).read()
    for prec in (1, 2, 3):
        for rm in (0, 1):
            sym = "_Z20hav_march_blk_kernelILi%dELi%dELi2EEv9MarchArgs" % (rm, prec)
            m = re.search(r"\.name:\s+" + sym + r"\b", text)
            assert m, sym
            meta = text[text.rfind("- .agpr_count", 0, m.start()):m.start() + 600]
            spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
            assert spills == 0 and int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0, (sym, spills)
            body = _kernel_body(text, sym)
            loops = _innermost_matrix_loops(body)
            assert len(loops) >= 2, (sym, loops)                 # the coarse sample loop and the loop over the new fine samples
            for a, b in loops:
                assert sum("v_mfma" in x for x in body[a:b]) >= 100, (sym, a, b)
                bad = [x.strip() for x in body[a:b] if re.match(r"\s+scratch_", x)]
                assert not bad, (sym, "scratch access inside a tile loop:", bad[:4])
    for sym in ("_Z20hav_march_blk_kernelILi0ELi2ELi2EEv9MarchArgs", "_Z20hav_march_blk_kernelILi1ELi2ELi2EEv9MarchArgs",
                "_Z20hav_march_blk_kernelILi1ELi1ELi2EEv9MarchArgs", "_Z20hav_march_blk_kernelILi1ELi3ELi2EEv9MarchArgs"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_war_check.py"), asm["hav_render"], sym, "8", "11"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert re.search(r"\b[1-9]\d* v_mfma instructions, \d+ early writes of an A/B operand", r.stdout), r.stdout[-500:]
        assert "0 touches of a matrix result inside 11 wait states" in r.stdout, r.stdout[-500:]
