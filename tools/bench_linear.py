#!/usr/bin/env python3
"""`linear` node op at the model's block shapes: fp16-split path (pre-split pass + GEMM) vs the exact-fp32 MFMA GEMM."""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip
DEV = 'cuda:0'


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for c, t in ((600, 1000), (800, 1000), (1000, 500), (1200, 250)):
    b = 64
    x = torch.randn(b, c, hip.round_up4(t), device=DEV)
    w, bias = torch.randn(c, c, device=DEV) * 0.03, torch.randn(c, device=DEV)
    y = torch.empty_like(x)
    packed, ws = hip.pack_pointwise_weights(w), hip.pointwise_workspace(b, c, x.shape[2], DEV)
    w3 = w.unsqueeze(-1).contiguous()
    t16 = timeit(lambda: hip.linear_fused_packed(x, t, packed, c, bias, (), y, ws))
    t32 = timeit(lambda: hip.dense_conv1d_fused(x, t, w3, bias, (), y, 1))
    gf = 2.0 * b * t * c * c / 1e9
    print(f'linear C={c:5d} T={t:5d}: f16x2 {t16 * 1e3:7.1f} us ({gf / t16:6.1f} TF)   fp32 {t32 * 1e3:7.1f} us ({gf / t32:6.1f} TF)', flush=True)
