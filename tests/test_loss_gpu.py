"""GPU parity of the ArcFace identity loss (photoverse_amd/loss.py) against the oracle restatement of models/loss.py + models/arcface_resnet.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _make_pair(smooth):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.arcface_ref import ArcFaceResNet18Ref, FaceLossRef
    from photoverse_amd.loss import ArcFaceResNet18, FaceLoss
    torch.manual_seed(11)
    ref_net = ArcFaceResNet18Ref().eval()
    g = torch.Generator().manual_seed(12)
    # calibrate the BatchNorm statistics on a batch of images (momentum 1: running stats = batch stats), as a trained network's are: with
    # the init values every image maps to nearly the same embedding (cos ~ 0.999) and d cos / d e2 becomes a difference of nearly equal
    # vectors that amplifies the fp16 rounding of the embeddings ~100x (measured 5e-2 on the image gradient in that regime)
    for m in ref_net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.momentum = 1.0
            m.weight.data.uniform_(0.7, 1.3, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)
        if isinstance(m, torch.nn.PReLU):
            m.weight.data.uniform_(0.1, 0.4, generator=g)
            if smooth:
                m.weight.data.fill_(1.0)          # PReLU(slope 1) = identity: no derivative jumps at 0
    ref_net.train()
    with torch.no_grad():
        ref_net(torch.randn(8, 1, 128, 128, generator=g) * torch.linspace(0.2, 1.5, 8).view(8, 1, 1, 1))
    ref_net.eval()
    net = ArcFaceResNet18()
    net.load_state_dict(ref_net.state_dict())
    return FaceLossRef(ref_net), FaceLoss("cuda", "arcface", model=net)


@pytest.fixture(scope="module")
def pair():
    return _make_pair(False)


@pytest.fixture(scope="module")
def smooth_pair():
    return _make_pair(True)


@pytest.mark.parametrize("H,normalize,maximize", [(96, False, True), (256, False, True), (128, True, True), (160, False, False)])
@pytest.mark.parametrize("smooth", [False, True])
def test_face_loss_value_and_image_gradient(pair, smooth_pair, H, normalize, maximize, smooth):
    """loss.py:64-78 and its gradient w.r.t. the generated image (what train.py:532-536 back-propagates): up-sampling (96 -> 128),
    down-sampling (256 -> 128), identity resize, the / 127.5 - 1 normalisation, both targets.

    Tolerances: the loss VALUE agrees to ~1e-4.  The image gradient of the real network agrees to 6-8e-2: the trunk has 17 PReLU layers with
    a derivative jump of 0.6-0.9 at zero plus a max-pool, and an fp16 forward flips the side of ~4e-4 of the units per layer relative to
    the fp32 oracle (sqrt(4e-4) * 0.75 ~ 1.5 % per layer, x sqrt(17) ~ 6 %) - the same effect the adapter test documents for LeakyReLU.
    ``smooth`` pins the plumbing itself: with every PReLU slope set to 1 (no kinks; the max-pool's arg-max flips remain: the whole gradient
    crosses that one layer, a flipped window moves its gradient to a neighbouring pixel) the same comparison gives 1.8-3.7e-2;
    each kernel of the chain is checked exactly in tests/test_hip_kernels.py::test_arcface_loss_pieces."""
    ref, hip = smooth_pair if smooth else pair
    g = torch.Generator().manual_seed(H)
    B = 2
    scale = 127.5 if normalize else 1.0
    x = (torch.rand(B, 3, H, H, generator=g) * 2 - 1) * scale + (scale if normalize else 0)
    # an unrelated second image: with x_gen close to x the embeddings nearly coincide and d cos / d e2 is a small difference of nearly equal
    # vectors, i.e. the fp16 rounding of the embeddings (5e-4) is amplified ~100x in the gradient (measured 5e-2 there)
    xg = ((torch.rand(B, 3, H, H, generator=g) * 2 - 1) * scale + (scale if normalize else 0)).roll(1, 0) * torch.linspace(0.2, 1.0, H).view(1, 1, H, 1)
    xr = xg.clone().requires_grad_()
    want = ref(x, xr, maximize=maximize, normalize=normalize)
    want.backward()
    loss, dimg = hip.loss_and_grad(x.cuda(), xg.cuda(), maximize=maximize, normalize=normalize)
    torch.cuda.synchronize()
    print(f"face loss H={H} smooth={smooth}: {loss.item():.5f} vs {want.item():.5f}; grad rel-L2 {rel_l2(dimg, xr.grad):.3e}")
    # measured: loss to 4 digits; image gradient 6.4-7.7e-2 (real trunk: PReLU / max-pool kinks under an fp16 forward), 1.8-3.7e-2 (smooth)
    assert loss.item() == pytest.approx(want.item(), rel=2e-3, abs=2e-4)
    assert rel_l2(dimg, xr.grad) < (5e-2 if smooth else 1e-1)
    assert float(hip(x.cuda(), xg.cuda(), maximize=maximize, normalize=normalize)) == pytest.approx(loss.item())


def test_face_loss_api(pair):
    from photoverse_amd.loss import FaceLoss
    with pytest.raises(NotImplementedError):
        FaceLoss("cuda", "facenet")
    _, hip = pair
    with pytest.raises(RuntimeError):
        hip.loss_and_grad(torch.zeros(1, 3, 128, 128), torch.zeros(1, 3, 128, 128))
