"""A FLOOR for one optimiser step of the persistent PPO update (`ppo_update_persistent_kernel`, policy.hip), so that
`us_per_step` has a denominator: the longest dependent path of a step -- layer chain, loss, dz, gradient tiles, hop 1,
slice sums, hop 2, norm, Adam -- priced with MI355X_MICROARCH.md's constants, from the ALGORITHM's minimum instruction
counts (what the step must execute whatever the compiler does); the ISA of the built kernel is walked beside it as a
cross-check (static MFMA / transcendental / LDS / exchange-word instruction counts: the model may never assume LESS
work than the algorithm's, and the ISA shows how much MORE the real stream carries).

Constants (MI355X_MICROARCH.md, "Per-instruction cycle constants", "Two waves per SIMD", "price list"):
  * one wave issues at most one instruction per ~4 cycles ("32 cyc/SIMD ~ 8 issue slots of ~4 cyc"); two waves on a SIMD
    issue alternately: a SIMD retires <= 1 VALU instruction per 2 cycles (`v_fma_f32` wave64: 2 cyc on SIMD-32);
  * `v_mfma_f32_16x16x4_f32`: 32 cycles of the SIMD's matrix pipe per instruction, 40 cycles dependent latency;
  * LDS: `ds_read_b32` ~50-64 cycles issue -> use; `ds_read_b128` 4 LDS cycles per wave-instruction;
  * one-way hand-off of 8-byte (value, sequence) words between CUs through the fabric: 0.8 us idle for 8 B, 1.0 us for
    <= 4 KB ("handoff-1to1"; by endpoint class unloaded -> unloaded 1.1, streaming -> streaming 2.9);
  * clock 2.4 GHz (max; the kernel occupies 19 of 256 CUs).
A phase's floor is max(matrix-pipe time of BOTH waves of the SIMD, VALU issue of both waves at 2 cycles, the critical
wave's own stream at 4 cycles per instruction + the dependent latencies it cannot overlap). Nothing in the model is
measured on the kernel itself; `profiles/r06_ppo_floor.md` sets the measured phase clocks and PMC instruction counts
beside it.

Usage: python tools/ppo_step_floor.py [obs_dim act_dim rows_per_minibatch]      (default 17 6 1024 = config P)
       python tools/ppo_step_floor.py --json ...                                 one JSON object (bench.py reads it)
CPU only."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

GHZ = 2.4
ISSUE_1WAVE = 4.0      # cycles per instruction of ONE wave
ISSUE_SIMD_VALU = 2.0  # cycles per wave64 VALU instruction on a SIMD (two waves alternating)
MFMA_ISSUE, MFMA_DEP = 32.0, 40.0
LDS_LAT = 64.0         # ds_read issue -> use (guide: ~50 for b32 alone; 64 with a b128 / busy LDS)
HOP_US = 0.8           # one-way 8-byte-word hand-off, idle endpoints (guide: 0.8 at 8 B ... 1.0 at 4 KB)
TANH_VALU, TANH_TRANS = 11, 2   # `fast_tanh`: 11 VALU + v_exp + v_rcp per value (policy.hip)


def model(D=17, A=6, rows=1024, discrete=False):
    """Per-phase floors in cycles for the workgroup form the host picks: several 64-row gradient workgroups (rows > 64), one
    (rows <= 64, `LOCAL`: no hops) or one with <= 16 rows (`SMALL`: one chain wave per SIMD)."""
    nblk = -(-rows // 64)
    local, small = nblk == 1, rows <= 16
    S1 = (D + 3) // 4                       # first layer's k steps of 4 inputs
    KT1 = -(-D // 16)                       # ... in K tiles of 16 (two accumulator chains per output tile)
    P = 2 * (32 * D + 32 + 32 * 32 + 32) + 32 * A + A + (0 if discrete else A) + 32 + 1
    npt = -(-P // 512)
    waves_per_simd = 1 if small else 2      # chain waves sharing a SIMD: (policy, q) and (value, q); SMALL: one
    ph = []

    def phase(name, mfma_pol, mfma_val, valu_pol, valu_val, dep_extra, note):
        pipe = (mfma_pol + (mfma_val if waves_per_simd == 2 else 0)) * MFMA_ISSUE
        valu = (valu_pol + (valu_val if waves_per_simd == 2 else 0)) * ISSUE_SIMD_VALU
        own = max(mfma_pol * MFMA_ISSUE, 0) + valu_pol * ISSUE_1WAVE + dep_extra   # the policy wave alone, nothing overlapped
        # (within one wave the VALU tail of a layer -- bias-free sums, tanh -- depends on the layer's last MFMAs: serial)
        ph.append(dict(name=name, cycles=max(pipe, valu, own), pipe=pipe, valu=valu, own=own, note=note))

    n1 = 2 * 4 * KT1                        # MFMAs of layer 1 per wave: 2 output tiles x 4 k steps x K tiles
    tanh = 8 * (TANH_VALU + TANH_TRANS) + 8  # 8 values per lane: sum of the two chains + tanh
    phase("fragments + per-row scalars", 0, 0, 6 + 4 * 2 * KT1 + 8 + 1 + 12, 2 + 4 * 2 * KT1 + 8, LDS_LAT,
          "LDS -> VGPR: loss scalars, W1 fragments, b1; Gaussian constants (exp, rcp, log + 8 lane picks)")
    phase("layer 1", n1, n1, 2 * KT1 * 4 + tanh + 8, 2 * KT1 * 4 + tanh + 8, LDS_LAT + (MFMA_DEP - MFMA_ISSUE),
          "x reads, 2 tiles x K MFMAs (two chains each), tanh of 8 values per lane, 8 tile writes; W2 fragment reads ride in the MFMA gaps")
    phase("layer 2", 16, 16, tanh + 8, tanh + 8, MFMA_DEP - MFMA_ISSUE, "16 MFMAs, tanh, tile writes; head fragments in the gaps")
    phase("heads", 8, 8, 4, 4, MFMA_DEP - MFMA_ISSUE, "8 MFMAs in two chains, 4 sums")
    loss_pol = (24 if not discrete else 40) + 8 + 8 + 12 + 3 + 4 + 8 + 3 + 5 + 36 + 14 + 24
    phase("loss (policy waves)", 0, 0, loss_pol, 12, LDS_LAT,
          "log-prob / entropy over 4 actions per lane, two 4-lane sums, advantage normalisation (one IEEE division), ratio, "
          "clip, d log-prob, head gradients + statistics rows written, backward fragments requested")
    phase("dz2, dz1", 8 + 16, 16, 24 + 32 + 16, 24 + 32 + 16, 2 * (MFMA_DEP - MFMA_ISSUE),
          "policy: 8 + 16 MFMAs, value: VALU outer product + 16 MFMAs; tanh' products; tile writes")
    # gradient tiles: 16 x 16 x 64 tiles, 16 MFMAs each; per SIMD <= 64 MFMAs (several workgroups: one-row / one-column
    # tiles are VALU dots), operand reads 8 ds_read_b128 per tile
    few = rows <= 16
    tile_mfma = 16 if few else 64
    phase("gradient tiles", tile_mfma // (1 if small else 2), tile_mfma // 2, 30, 30, LDS_LAT + 8 * 4 * 3,
          "dW2 / dW1 / head tiles contracted over the rows (<= 64 MFMAs per SIMD; <= 16 rows: four row steps), bias column "
          "sums, 8-byte slab words stored")
    chain = sum(p["cycles"] for p in ph)
    out = dict(config=dict(obs_dim=D, act_dim=A, rows_per_minibatch=rows, gradient_workgroups=nblk, parameters=P,
                           params_per_thread=npt, form="SMALL" if small else ("LOCAL" if local else "several workgroups")),
               phases=ph, chain_cycles=chain)
    us = lambda c: c / (GHZ * 1e3)
    rest = []
    rest.append(("row prefetch issue (next minibatch)", us(40 * ISSUE_1WAVE), "~40 instructions of address arithmetic + LDS-direct loads"))
    if not local:
        rest.append(("hop 1: slab words -> slice owner", HOP_US, "one one-way trip through the fabric (fire-and-forget 8-byte words)"))
        rest.append(("slice sums + publish", us(LDS_LAT * 2 + (nblk + 12) * ISSUE_1WAVE + 40), "LDS scratch, block barrier, nblk adds, words stored"))
        rest.append(("hop 2: sum vector -> every workgroup", HOP_US, "second one-way trip (the next minibatch is staged under it)"))
    else:
        rest.append(("gradient image LDS -> registers, barrier", us(LDS_LAT + npt * ISSUE_1WAVE + 40), "one gradient workgroup: nothing leaves the CU"))
    rest.append(("global norm", us(npt * ISSUE_1WAVE + 6 * 8 + LDS_LAT + 40 + 20 * ISSUE_1WAVE), "squares, DPP wave sum, one barrier, sqrt, clip coefficient (one division)"))
    adam = 2 * npt + npt * 14 + 2 * npt
    rest.append(("Adam", us(max(adam * ISSUE_1WAVE, 2 * adam * ISSUE_SIMD_VALU) + LDS_LAT), f"{npt} parameters per thread x 14 VALU/transcendental, LDS reads and writes"))
    out["rest"] = [dict(name=n, us=u, note=t) for n, u, t in rest]
    out["chain_us"] = us(chain)
    out["floor_us"] = us(chain) + sum(u for _, u, _ in rest)
    return out


def isa_counts(D, rows, npt_kernel=8):
    """Static instruction counts of the production instantiation in the built library (cross-check only)."""
    try:
        import isa_pattern
    except Exception:
        return None
    nblk = -(-rows // 64)
    ks1 = 8 if D <= 32 else 16
    sub = f"ppo_update_persistent_kernel<{npt_kernel}, false, {ks1}, {'true' if nblk == 1 else 'false'}, false, {'true' if rows <= 16 else 'false'}>"
    try:
        found = list(isa_pattern.disassemble(sub))
    except Exception:
        return None
    if not found:
        return None
    name, lines = found[0]
    cnt = lambda pat: sum(1 for ln in lines if re.search(pat, ln))
    return dict(kernel=sub, instructions=len(lines), mfma_16x16x4=cnt(r"v_mfma_f32_16x16x4"), mfma_32x32x2=cnt(r"v_mfma_f32_32x32x2"),
                transcendental=cnt(r"v_(exp|rcp|sqrt|rsq|log)_f32"), ds_read_b128=cnt(r"ds_read_b128"), ds_read_b32=cnt(r"ds_read_b32|ds_read2_b32"),
                ds_write=cnt(r"ds_write"), global_load_sc1=cnt(r"global_load_dwordx2.*sc1"), global_store_sc1=cnt(r"global_store_dwordx2.*sc1"),
                global_load_lds=cnt(r"global_load_lds"), s_barrier=cnt(r"s_barrier"), valu=cnt(r"^\s*v_"), salu=cnt(r"^\s*s_"))


def main():
    args = [a for a in sys.argv[1:] if a != "--json"]
    D, A, rows = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (17, 6, 1024)
    m = model(D, A, rows)
    if "--json" in sys.argv:
        print(json.dumps(dict(floor_us=m["floor_us"], chain_us=m["chain_us"], config=m["config"])))
        return
    print(f"floor of one optimiser step, {m['config']}")
    print(f"{'phase':44s} {'floor clk':>9s} {'us':>6s}   matrix pipe / VALU issue (SIMD) / critical wave alone")
    for p in m["phases"]:
        print(f"  {p['name']:42s} {p['cycles']:9.0f} {p['cycles'] / GHZ / 1e3:6.2f}   {p['pipe']:.0f} / {p['valu']:.0f} / {p['own']:.0f}")
    print(f"  {'chain':42s} {m['chain_cycles']:9.0f} {m['chain_us']:6.2f}")
    for r in m["rest"]:
        print(f"  {r['name']:42s} {'':9s} {r['us']:6.2f}   {r['note']}")
    print(f"FLOOR {m['floor_us']:.2f} us per optimiser step")
    isa = isa_counts(D, rows, 8 if m['config']['parameters'] <= 4096 else 9)
    if isa:
        print("ISA of the built kernel (static counts, every path of every wave):", json.dumps(isa))
        need = sum(2 * (8 * (-(-D // 16))) for _ in (0,)) // 2 + 16 + 8 + 8 + 16   # MFMAs of the chain a policy wave runs once per step
        assert isa["mfma_16x16x4"] >= need, (isa["mfma_16x16x4"], need)


if __name__ == "__main__":
    main()
