"""GPU tests of the bf16 MFMA dense kernel against a float64 reference of the same bf16 inputs."""
import pytest
import torch

from snipper_amd.dense import linear_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # (M, K, N)        ragged M and N on purpose
    (1, 64, 4), (130, 64, 20), (257, 128, 132), (1000, 384, 96), (4099, 384, 192),
    (2050, 1024, 384), (777, 384, 1024), (513, 2048, 512), (300, 256, 64),
]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("relu,res,bias", [(False, False, True), (True, True, True), (True, False, False)])
def test_linear_bf16_matches_reference(M, K, N, relu, res, bias):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g).to(torch.bfloat16) if res else None
    ref = x.double() @ w.double().T
    if b is not None:
        ref = ref + b.double()
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    got = linear_bf16(x.to(DEV), w.to(DEV), b.to(DEV) if bias else None, r.to(DEV) if res else None, relu)
    assert got.dtype == torch.bfloat16 and got.shape == (M, N)
    # f32 accumulation + one bf16 rounding of the result: half an ulp of bf16 = 2^-9 relative
    torch.testing.assert_close(got.double().cpu(), ref, rtol=2 ** -8, atol=2e-2)


def test_linear_bf16_strided_rows_and_batch_dims():
    g = torch.Generator().manual_seed(0)
    big = torch.randn(3, 50, 256, generator=g).to(torch.bfloat16).to(DEV)
    x = big[..., :128]                       # rows strided by 256, last dim contiguous
    w = torch.randn(64, 128, generator=g).to(torch.bfloat16).to(DEV)
    got = linear_bf16(x, w)
    ref = (x.double() @ w.double().T)
    assert got.shape == (3, 50, 64)
    torch.testing.assert_close(got.double(), ref, rtol=2 ** -8, atol=2e-2)
    with pytest.raises(RuntimeError):
        linear_bf16(big[..., :100], torch.zeros(64, 100, dtype=torch.bfloat16, device=DEV))   # K % 64 != 0


def test_frozen_bottleneck_on_hip_matches_miopen():
    """bf16 NHWC, no grad: the Bottleneck's 1x1 convolutions take the MFMA kernel; compare with the
    same block forced through F.conv2d."""
    import snipper_amd.backbone as bb
    torch.manual_seed(0)
    blk = bb.Bottleneck(256, 64).to(DEV)
    for m in blk.modules():
        if isinstance(m, bb.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 256, 20, 24, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert bb._hip_pointwise_ok(x, blk.conv1, blk.conv1.weight)
        got = blk(x)
        saved = bb._hip_pointwise_ok
        bb._hip_pointwise_ok = lambda *a: False
        try:
            ref = blk(x)
        finally:
            bb._hip_pointwise_ok = saved
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(got.float(), ref.float(), rtol=3e-2, atol=3e-2)
