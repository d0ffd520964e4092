"""GPU parity of the ArcFace identity loss (photoverse_amd/loss.py) against the oracle restatement of models/loss.py + models/arcface_resnet.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def pair():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.arcface_ref import ArcFaceResNet18Ref, FaceLossRef
    from photoverse_amd.loss import ArcFaceResNet18, FaceLoss
    torch.manual_seed(11)
    ref_net = ArcFaceResNet18Ref().eval()
    g = torch.Generator().manual_seed(12)
    for m in ref_net.modules():                       # non-trivial eval-mode statistics / affine parameters / slopes
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.2, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
            m.weight.data.uniform_(0.7, 1.3, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)
        if isinstance(m, torch.nn.PReLU):
            m.weight.data.uniform_(0.1, 0.4, generator=g)
    net = ArcFaceResNet18()
    net.load_state_dict(ref_net.state_dict())
    return FaceLossRef(ref_net), FaceLoss("cuda", "arcface", model=net)


@pytest.mark.parametrize("H,normalize,maximize", [(96, False, True), (256, False, True), (128, True, True), (160, False, False)])
def test_face_loss_value_and_image_gradient(pair, H, normalize, maximize):
    """loss.py:64-78 and its gradient w.r.t. the generated image (what train.py:532-536 back-propagates): up-sampling (96 -> 128),
    down-sampling (256 -> 128), identity resize, the / 127.5 - 1 normalisation, both targets."""
    ref, hip = pair
    g = torch.Generator().manual_seed(H)
    B = 2
    scale = 127.5 if normalize else 1.0
    x = (torch.rand(B, 3, H, H, generator=g) * 2 - 1) * scale + (scale if normalize else 0)
    xg = (x + 0.6 * scale * torch.randn(B, 3, H, H, generator=g)).clamp(-scale if not normalize else 0, scale if not normalize else 255)
    xr = xg.clone().requires_grad_()
    want = ref(x, xr, maximize=maximize, normalize=normalize)
    want.backward()
    loss, dimg = hip.loss_and_grad(x.cuda(), xg.cuda(), maximize=maximize, normalize=normalize)
    torch.cuda.synchronize()
    print(f"face loss H={H}: {loss.item():.5f} vs {want.item():.5f}; grad rel-L2 {rel_l2(dimg, xr.grad):.3e}")
    assert loss.item() == pytest.approx(want.item(), rel=2e-2, abs=2e-3)
    assert rel_l2(dimg, xr.grad) < 5e-2
    assert float(hip(x.cuda(), xg.cuda(), maximize=maximize, normalize=normalize)) == pytest.approx(loss.item())


def test_face_loss_api(pair):
    from photoverse_amd.loss import FaceLoss
    with pytest.raises(NotImplementedError):
        FaceLoss("cuda", "facenet")
    _, hip = pair
    with pytest.raises(RuntimeError):
        hip.loss_and_grad(torch.zeros(1, 3, 128, 128), torch.zeros(1, 3, 128, 128))
