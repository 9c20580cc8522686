"""Developer tool: the mask head's launch alone (se_head_linear_pre_f32) at the configs[3] bench shape, event-timed; SE_AMD_LIB selects a tagged build
(ablations: SE_AMD_BUILD_TAG=hablN SE_AMD_EXTRA_DEFINES=-DSE_HEAD_ABL=N python build.py)."""
import sys
import torch
sys.path.insert(0, '.')
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

B, F, D, N = 256, 1001, int(sys.argv[1]) if len(sys.argv) > 1 else 120, 201
dev = torch.device('cuda:0')
lib = L.load()
torch.manual_seed(0)
feats = torch.randn(B, F, D, device=dev)
lin = torch.rand(B, F, N, device=dev)
W, bias = torch.randn(N, D, device=dev) * 0.1, torch.randn(N, device=dev)
w3 = torch.empty(lib.se_head_w3_bytes(N, D), device=dev, dtype=torch.uint8)
L.check(lib.se_head_split_weights_f32(L.ptr(W), N, D, L.ptr(w3), L.stream()), 'split')
stats = torch.empty(B, D, 2, device=dev)
L.check(lib.se_head_colstats_f32(L.ptr(feats), B, F, D, 1e-6, L.ptr(stats), L.stream()), 'colstats')
out = torch.empty(B, F, N, device=dev)


def run():
    L.check(lib.se_head_linear_pre_f32(L.ptr(feats), L.ptr(w3), L.ptr(bias), L.ptr(lin), L.ptr(stats), B, F, D, N, 2, L.ptr(out), None, L.stream()), 'head')


for _ in range(5):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
gb = 4.0 * B * F * (D + 2 * N) / 1e9
print(f'head D={D}: {min(ts):7.1f} us (min) {sorted(ts)[len(ts) // 2]:7.1f} us (median)  {gb / min(ts) * 1e3:6.2f} TB/s')
