#!/usr/bin/env python3
"""LayerNorm + downsample conv at the model's three LayerNorm-fed convolutions: materialised LayerNorm (+ max|y|) followed by
the staging fp16 conv, vs LayerNorm written as the pre-split image followed by the image-gathering conv."""
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
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for c, cout, t, s in ((600, 800, 1000, 1), (800, 1000, 1000, 2), (1000, 1200, 500, 2)):
    b = BATCH
    x = torch.randn(b, c, hip.round_up4(t), device=DEV)
    g, be = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2
    w, bias = torch.randn(cout, c, 8, device=DEV) * 0.02, torch.randn(cout, device=DEV)
    t_out = (t + s - 1) // s
    y = torch.empty(b, cout, hip.round_up4(t_out), device=DEV)
    normed, amax = torch.empty_like(x), torch.empty(b, device=DEV)
    packed = hip.pack_dense_weights(w, s, 'f16x2')
    stats, bound, image = torch.empty(b, 2, x.shape[2], device=DEV), torch.empty(b, device=DEV), hip.split_image(b, c, x.shape[2], DEV)
    ln_a = timeit(lambda: hip.layernorm_channels(x, g, be, normed, t, 1e-3, amax))
    cv_a = timeit(lambda: hip.dense_conv1d_fused_packed(normed, t, packed, cout, 8, bias, (), y, s, scheme='f16x2', x_absmax=amax))
    ln_b = timeit(lambda: hip.layernorm_split_image(x, g, be, stats, bound, image, t, 1e-3))
    cv_b = timeit(lambda: hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, x.shape[2], packed, cout, 8, bias, y, s))
    packed160 = hip.pack_dense_weights(w, s, 'f16x2', row_tile=160)
    cv_c = timeit(lambda: hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, x.shape[2], packed160, cout, 8, bias, y, s, row_tile=160))
    print(f'B={b} {c}->{cout} T={t} s={s}: LN {ln_a:6.1f} + conv {cv_a:7.1f} = {ln_a + cv_a:7.1f} us   |   LN-split {ln_b:6.1f} + conv-img {cv_b:7.1f} = {ln_b + cv_b:7.1f} us   |   conv-img, 160-row tiles {cv_c:7.1f} us', flush=True)
