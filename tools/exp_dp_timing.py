"""Where does the short scorer's kernel time go?  Runs the C2 workload against the DEBUG build of the library
(make -C frizbee_amd/csrc timing -> libfrizbee_hip_timing.so: per-wave s_memrealtime / s_memtime stamps at entry and exit of
k2b_dp_short) and reduces the stamps of the last launch: dispatch ramp (first -> last wave start), resident time per wave, tail
(first -> last wave end), and the effective shader clock = s_memtime ticks / s_memrealtime ticks * 100 MHz.
Usage on the GPU box: FRIZBEE_HIP_LIB=frizbee_amd/libfrizbee_hip_timing.so python tools/exp_dp_timing.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F

dev = torch.device("cuda", 0)
n = 10_000_000
flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32)
for typos in (0, 2):
    m = F.Matcher("deadbe", F.Config(max_typos=typos, pf_lanes=64, sw_lanes=64))
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(5):
        m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize()
    buf = np.zeros(6 * 8192, dtype=np.uint64)
    rc = F.lib().fzb_debug_dp_timing(C.c_void_p(buf.ctypes.data))
    assert rc == 0, rc
    t = buf.reshape(-1, 6)
    t = t[t[:, 1] > 0]
    rt0, rt1, ck0, ck1 = (t[:, i].astype(np.float64) for i in range(4))
    base = rt0.min()
    us = lambda ticks: ticks / 100.0  # 100 MHz -> microseconds
    res = {"max_typos": typos, "waves": int(len(t)), "matches": int(cnt[0].item()),
           "first_to_last_wave_start_us": us(rt0.max() - base), "start_p50_us": us(np.median(rt0) - base), "start_p99_us": us(np.percentile(rt0, 99) - base),
           "resident_us_min_p50_max": [us((rt1 - rt0).min()), us(np.median(rt1 - rt0)), us((rt1 - rt0).max())],
           "first_wave_end_us": us(rt1.min() - base), "end_p50_us": us(np.median(rt1) - base), "last_wave_end_us": us(rt1.max() - base),
           "effective_clock_GHz_p50": float(np.median((ck1 - ck0) / np.maximum(rt1 - rt0, 1.0) * 0.1)),
           "shader_cycles_resident_p50": float(np.median(ck1 - ck0))}
    print(json.dumps(res), flush=True)
    # placement: HW_ID = wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]
    hw, xcc = t[:, 4].astype(np.int64), t[:, 5].astype(np.int64) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = us(rt1 - rt0)
    endt = us(rt1 - base)
    cus = np.unique(key)
    per_cu = np.array([(key == k).sum() for k in cus])
    per_simd = np.array([((key == k) & (simd == sd)).sum() for k in cus for sd in range(4)])
    cu_end = np.array([endt[key == k].max() for k in cus])
    print(json.dumps({"distinct_cus": int(len(cus)), "waves_per_cu_hist": np.bincount(per_cu).tolist(), "waves_per_simd_hist": np.bincount(per_simd).tolist(),
                      "cu_last_end_us_min_p50_max": [float(cu_end.min()), float(np.median(cu_end)), float(cu_end.max())],
                      "per_xcc_last_end_us": [float(endt[xcc == xx].max()) if (xcc == xx).any() else None for xx in range(8)],
                      "per_xcc_clock_GHz": [float(np.median(((ck1 - ck0) / np.maximum(rt1 - rt0, 1.0) * 0.1)[xcc == xx])) if (xcc == xx).any() else None for xx in range(8)],
                      "end_by_waves_on_cu": {int(c): float(np.median(cu_end[per_cu == c])) for c in np.unique(per_cu)},
                      "iterations_of_wave_vs_dur": [float(np.median(dur[: 3717])), float(np.median(dur[3717:]))] if typos == 0 else None}), flush=True)
