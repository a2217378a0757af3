import sys, torch
sys.path.insert(0, '/root/repo')
from outlier_suppression_amd import ops
dev = torch.device("cuda:0")
n = (1 << 31) + 12345          # > 2^31 elements: 8.6 GB
x = torch.empty(n, device=dev)
for i in range(0, n, 1 << 28):
    x[i:i + (1 << 28)].normal_()
x[n - 3] = 77.0; x[5] = -99.0
s = torch.tensor([0.05], device=dev); z = torch.tensor([31], dtype=torch.int32, device=dev)
y = ops.fake_quant_per_tensor(x, s, z, 0, 63)
for lo in (0, (1 << 31) - 1000, n - 5000):
    xs = x[lo:lo + 4096]
    ref = (torch.clamp((xs / 0.05).round() + 31, 0, 63) - 31) * 0.05
    assert torch.equal(y[lo:lo + 4096], ref), lo
mn = torch.tensor(float("inf"), device=dev); mx = torch.tensor(float("-inf"), device=dev)
ops.observe_flat(x, ops.UPDATE_RUNNING, 0, mn, mx, 0, 63, False)
print("fq ok; min/max", mn.item(), mx.item())
assert mn.item() == -99.0 and mx.item() == 77.0
