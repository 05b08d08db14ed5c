#!/usr/bin/env python3
"""Diagnostic (EXP_W4_TIMELINE build only): phases of a 4-wave window-kernel workgroup from s_memtime stamps, and the
shader clock the chip holds inside the K loop (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md
'DVFS give-back' item 6), after two seconds of back-to-back launches on random data.
usage: tools/exp_build.sh w4tl -DEXP_W4_TIMELINE; ONEDC_HIP_F16_SO=tools/exp/libonedc_hip_w4tl.so timeline_win4.py B H W Cin Cout
       [--res] (residual read in the epilogue)  [--gn] (GroupNorm + SiLU on load: MODE 1)  [--bf] (bf16 operands)
       [--epilogue] (library built with -DEXP_W4_TIMELINE -DEXP_W4_EPI_TL: slots 10 .. 13 are the epilogue's stamps, round 6)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

B, H, W, Cin, Cout = map(int, sys.argv[1:6])
FLAGS = set(sys.argv[6:])
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
y = K.new_view(B, H, W, Cout, DEV)
blocks = (Cout + 127) // 128 * (H // 16) * (W // 16) * B
dbg = torch.zeros(blocks * 4 * 16, dtype=torch.int64, device=DEV)


class FakeStats:
    t = dbg
    gran = 4

    def mark(self, *a):
        pass


y.stats = FakeStats()
res = K.new_view(B, H, W, Cout, DEV) if "--res" in FLAGS else None
if res is not None:
    res.t.copy_(torch.randn(res.t.numel(), generator=g, device=DEV).to(K.a16()))
gn = w_bf = None
if "--gn" in FLAGS:
    from onedc_amd.layers import Workspace
    gn = (torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 1e-6, True, Workspace(torch.device(DEV)))
    if "--bf" in FLAGS:
        w_bf = w.to(torch.bfloat16) if w.dtype != torch.bfloat16 else w
t_end = time.time() + 2.0
n = 0
while time.time() < t_end:
    for _ in range(20):
        K.conv_gemm(x, w, None, y, KH=3, tile=131, stats=True, res=res, gn=gn, w_bf16=w_bf)
    torch.cuda.synchronize()
    n += 20
d = dbg.view(blocks, 4, 16).cpu().numpy().astype(np.int64)
t0, t1, t2, t3 = d[:, :, 1].min(1), d[:, :, 2].max(1), d[:, :, 3].max(1), d[:, :, 4].max(1)
clk = (d[:, :, 3] - d[:, :, 2]) / np.maximum(d[:, :, 6] - d[:, :, 5], 1) * 100.0      # MHz
print(f"{blocks} workgroups, {n} launches; s_memtime ticks per workgroup (min start .. max end over its 4 waves):")
print(f"  prologue (entry -> first fragments) {np.mean(t1 - t0):9.0f}")
pa, pb, pc, pd = (np.mean(d[:, :, i] - d[:, :, 1]) for i in (7, 8, 9, 10))
if "--epilogue" not in FLAGS:
    q0, q1, q2, q3 = (np.mean(d[:, :, i] - d[:, :, 1]) for i in (11, 12, 13, 14))
    print(f"    of it (per wave, from entry): tile decoded {q0:.0f}, first weights issued {q1:.0f}, window offsets {q2:.0f}, window issued {q3:.0f},")
print(f"      every DMA issued {pa:.0f}, set-up arithmetic done {pb:.0f}, own DMA landed {pc:.0f}, barrier {pd:.0f}")
print(f"  K loop                              {np.mean(t2 - t1):9.0f}   ({(Cin // 32) * 9} K-tiles: {np.mean(t2 - t1) / ((Cin // 32) * 9):.0f} per K-tile)")
print(f"  epilogue                            {np.mean(t3 - t2):9.0f}")
if "--epilogue" in FLAGS:
    # per WAVE, from the end of its own K loop (tl[2] = column 3): 10 -> col 11, 12 -> col 13, 11 -> col 12, 3 -> col 4, 13 -> col 14
    e = {name: d[:, :, col] - d[:, :, 3] for name, col in (("barrier", 11), ("res03", 13), ("arith", 12), ("issued", 4), ("drained", 14))}
    seg = [("K loop end -> bias / residual rows 0-3 issued, lgkmcnt(0), barrier passed (staging rows free)", e["barrier"]),
           ("  -> residual rows 0 .. 3 landed (forced vmcnt(0); without --res: the bias loads)", e["res03"] - e["barrier"]),
           ("  -> 128 outputs per lane finished, rows 4 .. 7 of the residual waited for, LDS transpose written", e["arith"] - e["res03"]),
           ("  -> 16 line stores read back from LDS and issued", e["issued"] - e["arith"]),
           ("  -> stores drained (vmcnt(0))", e["drained"] - e["issued"])]
    tot = np.mean(e["drained"])
    print(f"  epilogue of ONE WAVE, s_memtime ticks from the end of its K loop (mean over all waves; p10 / p90), total {tot:.0f}:")
    for name, v in seg:
        print(f"    {np.mean(v):8.0f}  ({np.percentile(v, 10):6.0f} / {np.percentile(v, 90):6.0f})  {100 * np.mean(v) / tot:5.1f} %  {name}")
    t3 = d[:, :, 14].max(1)                     # the workgroup ends when its stores have drained
    print(f"  whole workgroup incl. drain         {np.mean(t3 - t0):9.0f}")
print(f"  whole workgroup                     {np.mean(t3 - t0):9.0f}")
print(f"  shader clock inside the K loop      median {np.median(clk):7.0f} MHz  (p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f})")

# ---- who shares a CU, and are the two resident workgroups in phase? ---------------------------------------------------
hw = d[:, 0, 0]
xcc, hwid = hw >> 32, hw & 0xffffffff
wave_id, simd, cu, sh, se = hwid & 15, (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("wave slot (HW_ID.WAVE_ID) of wave 0 over all workgroups:", np.bincount(wave_id, minlength=4)[:8])
first = np.arange(blocks) < 512
print("  ... of launch indices < 256:", np.bincount(wave_id[:256], minlength=4)[:4], " 256..511:", np.bincount(wave_id[256:512], minlength=4)[:4])
print("distinct CUs:", len(np.unique(cu_key)))
# K-loop interval overlap between the two slots of a CU: for every workgroup, the fraction of its prologue + epilogue time
# during which the OTHER slot of its CU was inside a K loop
ks, ke = d[:, 0, 2], d[:, 0, 3]
st, en = d[:, 0, 1], d[:, 0, 4]
frac = []
for c in np.unique(cu_key)[:64]:
    idx = np.where(cu_key == c)[0]
    for s in (0, 1):
        me = idx[(wave_id[idx] & 1) == s]
        ot = idx[(wave_id[idx] & 1) != s]
        for i in me[2:-2]:
            oh = [(st[i], ks[i]), (ke[i], en[i])]
            tot = sum(b - a for a, b in oh)
            cov = 0
            for a, b in oh:
                for j in ot:
                    cov += max(0, min(b, ke[j]) - max(a, ks[j]))
            frac.append(cov / max(tot, 1))
print(f"fraction of a workgroup's prologue + epilogue during which the CU's other slot is in its K loop: {np.mean(frac):.2f}")

# ---- are the workgroups of DIFFERENT CUs in phase?  (chip-wide burstiness of the epilogue's stores) ----------------------
# s_memrealtime (100 MHz, one counter for the chip) at the end of the K loop of every workgroup of the last launch.  If
# all CUs run their tiles in lockstep, the epilogues -- 64 KB of stores each -- hit the memory system together: the
# concurrency an epilogue sees is then far above the uniform-phase expectation slots * E / T.
ke_rt = d[:, :, 6].max(1).astype(np.float64)                    # 10 ns ticks
ks_rt = d[:, :, 5].min(1).astype(np.float64)
mhz = float(np.median(clk))
E = float(np.mean(t3 - t2)) / mhz * 100.0                       # epilogue length in 10 ns ticks
T = float(np.mean(t3 - t0)) / mhz * 100.0                       # whole workgroup
order = np.sort(ke_rt)
mid = order + E / 2
conc = np.searchsorted(order, mid, side="right") - np.searchsorted(order, mid - E, side="right")
slots = 2 * len(np.unique(cu_key))
print(f"epilogue {E / 100:.2f} us of a {T / 100:.2f} us workgroup; workgroups in their epilogue at the same time: mean {conc.mean():.0f}, "
      f"p10 {np.percentile(conc, 10):.0f}, p90 {np.percentile(conc, 90):.0f}, max {conc.max()}  (uniform phases would give {slots * E / T:.0f} of {slots} slots)")
span = order[-1] - ks_rt.min()
print(f"launch spans {span / 100:.1f} us = {span / T:.1f} workgroup times for {blocks / slots:.1f} rounds")
hist, _ = np.histogram((order - order[0]) % T, bins=16, range=(0, T))
print("K-loop end times modulo the workgroup time, 16 bins (flat = desynchronised):", hist.tolist())

# ---- how long does a CU slot sit empty between two workgroups? -----------------------------------------------------------
# consecutive workgroups on the same (CU, wave slot): entry stamp of the next minus end stamp of the previous (s_memtime of one
# CU: one clock domain).  What a persistent loop over tiles could recover at most (before what its vmcnt ordering costs).
gaps = []
for c in np.unique(cu_key):
    idx = np.where(cu_key == c)[0]
    for s in (0, 1):
        me = idx[(wave_id[idx] & 1) == s]
        if len(me) < 3:
            continue
        me = me[np.argsort(st[me])]
        g = st[me][1:] - en[me][:-1]
        gaps.extend(g[(g > -1000) & (g < 100000)].tolist())
gaps = np.asarray(gaps, dtype=np.float64)
print(f"empty slot time between consecutive workgroups of a CU slot: median {np.median(gaps):.0f} cycles, mean {gaps.mean():.0f}, "
      f"p90 {np.percentile(gaps, 90):.0f}  ({100 * gaps.mean() / (gaps.mean() + np.mean(t3 - t0)):.1f} % of a slot's time)")
