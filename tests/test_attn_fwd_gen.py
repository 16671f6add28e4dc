"""not gpu: the generated attention forward (tools/gen_attn_fwd.py -> kfunca_amd/csrc/device/attn_fwd_w4.inc).

The tile body is one hand-placed instruction stream; what hipcc would otherwise do for it the generator does itself, and this test
holds it to that: (1) the address maps of the LDS image - what the DMA pieces write, what the K row reads and the transposed V reads
deliver as MFMA operands - agree in a pure-Python model (`selftest`); (2) every variant of the emitted stream keeps the distances the
hardware does not interlock (`check`: MFMA result -> VALU, VALU -> MFMA operand, LDS read -> use behind a wait); (3) the committed .inc is
what the generator writes today (nobody edited one without the other)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def test_address_maps_and_hazard_distances():
    import gen_attn_fwd as G
    assert G.selftest()
    for f16 in (False, True):
        for mut in (False, True):
            for scaled in (False, True):   # exact f32 scores (default) | the query scaled and rounded once per pass (KF_ATTN_SCALED_OPERANDS)
                g = G.Gen(f16, mut, scaled=scaled).build()
                assert G.check(g.out) == []
                variants = {i.text[2:-4] for i in g.out if i.kind == "label" and i.text.endswith("_%=:")}
                assert {"steady", "first", "masked", "firstmasked", "drain", "idle"} <= variants and ("steadydrop" in variants) == mut
                assert sum(1 for i in g.out if i.kind == "mfma") == (64 + 48 + 56 + 40 + 16 + (64 if mut else 0))  # MFMAs per variant: steady, first, masked, firstmasked, drain
                steady = [i for i in G.gap_table(g.out, "steady")]
                assert sum(r["n"] for r in steady) > 0


def test_committed_inc_is_the_generators_output(tmp_path):
    out = tmp_path / "w4.inc"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_fwd.py"), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_text() == (ROOT / "kfunca_amd" / "csrc" / "device" / "attn_fwd_w4.inc").read_text(), "run tools/gen_attn_fwd.py"


def test_dkv_generator_address_maps_hazards_and_freshness(tmp_path):
    """The same three guarantees for the generated dK / dV pass (tools/gen_attn_dkv.py -> attn_dkv_w4.inc): tile-image model, hazard
    distances in every variant of the slice body (both dtypes, mutation build, with and without the dS stores), committed stream ==
    generator output."""
    import gen_attn_dkv as G
    assert G.selftest()
    for f16 in (False, True):
        for mut in (False, True):
            for ds in (True, False):
                for scaled in (False, True):   # exact f32 scores (default) | K scaled and rounded once per block
                    g = G.Gen(f16, mut, ds, scaled=scaled).build()
                    assert G.check(g.out) == []
                    assert sum(1 for i in g.out if i.kind == "mfma") == 16 + 64 + 64 + 32 + (64 if mut else 0)   # accumulator clearing, steady, diag1, diag0 (one sub-block), drop
                    stores = sum(1 for i in g.out if "global_store_dwordx4" in i.text)
                    assert stores == ((4 + 4 + 2 + (4 if mut else 0)) if ds else 0)
    out = tmp_path / "dkv.inc"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_dkv.py"), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_text() == (ROOT / "kfunca_amd" / "csrc" / "device" / "attn_dkv_w4.inc").read_text(), "run tools/gen_attn_dkv.py"


def test_checkers_reject_a_transcendental_read_one_wait_state_later():
    """gfx950 does not interlock a transcendental's result against the very next instruction (measured round 4, tools/scratch/trans_hazard.hip:
    `v_exp; v_add` as neighbours leave the old value in the lanes with (lane & 4) == 0; one wait state is what the hardware needs). The
    generators keep one in reserve: both checkers must flag an adjacent consumer and the `s_nop 0` form, and accept two wait states or a
    real instruction in between; the forward's stream, built WITHOUT its repair pass,
    must trip the rule (the drain iteration has gaps without MFMAs)."""
    import gen_attn_dkv as D
    import gen_attn_fwd as G
    Ins, V = G.Ins, G.V
    exp = lambda: Ins("v_exp_f32 v1, v1", "trans", V(1), V(1))       # noqa: E731
    use = lambda: Ins("v_add_f32 v2, v2, v1", "valu", V(2) + V(1), V(2))  # noqa: E731
    other = lambda: Ins("v_mov_b32 v3, v4", "valu", V(4), V(3))      # noqa: E731
    nop = lambda n: Ins(f"s_nop {n}", "salu")                        # noqa: E731
    for mod, label in ((G, "L_steady_%=:"), (D, "L_steady_%=:")):
        body = lambda mid: [Ins(label, "label"), exp()] + mid + [use(), Ins("L_epilogue_%=:", "label")]  # noqa: E731
        assert any("transcendental" in p for p in mod.check(body([])))
        assert any("transcendental" in p for p in mod.check(body([nop(0)])))
        assert not any("transcendental" in p for p in mod.check(body([nop(1)])))
        assert not any("transcendental" in p for p in mod.check(body([other()])))
    g = G.Gen()
    g.fix_trans_use = lambda: None
    assert any("transcendental" in p for p in G.check(g.build().out))


def test_head_size_64_streams_hazards_counts_and_wait_fields():
    """Round 5: both generators at D = 64 (the reference's second fast head size). Hazard distances in every variant, the MFMA counts of
    the half-size tile / slice (32 per steady body), and - the regression of the round - no s_waitcnt field beyond what the hardware has:
    lgkmcnt is 4 bits wide (the first D = 64 dK/dV stream asked for lgkmcnt(23): the assembler refused it, the old library stayed in
    place, and nothing said so)."""
    import re

    import gen_attn_dkv as D
    import gen_attn_fwd as G
    for f16 in (False, True):
        for mut in (False, True):
            g = G.Gen(f16, mut, D=64).build()
            assert G.check(g.out) == []
            assert sum(1 for i in g.out if i.kind == "mfma") == (32 + 24 + 28 + 20 + 8 + (32 if mut else 0))   # steady, first, masked, firstmasked, drain (+ steadydrop)
            for ds in (True, False):
                d = D.Gen(f16, mut, ds, D=64).build()
                assert D.check(d.out) == []
                assert sum(1 for i in d.out if i.kind == "mfma") == 8 + 32 + 32 + 16 + (32 if mut else 0)        # accumulator clearing, steady, diag1, diag0, drop
    for d in (D.Gen(False, D=64).build(), D.Gen(False, D=128).build(), D.Gen(True, scaled=True).build()):
        for i in d.out:
            for field, width in (("lgkmcnt", 15), ("vmcnt", 63)):
                for m in re.finditer(field + r"\((\d+)\)", i.text):
                    assert int(m.group(1)) <= width, i.text
    inc = (ROOT / "kfunca_amd" / "csrc" / "device" / "attn_dkv_w4.inc").read_text() + (ROOT / "kfunca_amd" / "csrc" / "device" / "attn_fwd_w4.inc").read_text()
    for name in ("KF_DKV_W4_D64_ASM_BF16_DS", "KF_DKV_W4_D64_ASM_F16_NODS", "KF_FWD_W4_D64_ASM_BF16", "KF_FWD_W4_D64_ASM_F16"):
        assert f"#define {name} " in inc, name
