"""TEST INFRASTRUCTURE (oracle): CPU fp32 restatement of the reference's ArcFace identity loss.

* ``ArcFaceResNet18Ref`` - ``/root/reference/models/arcface_resnet.py:12-133``: IR-ResNet18 without SE blocks (``use_se=False``,
  ``:127-128``): conv1(1->64) + BN + PReLU + MaxPool(2), four stages of two IRBlocks (BN -> conv3x3 -> BN -> PReLU -> conv3x3(stride) ->
  BN -> + residual (1x1 stride conv + BN where the shape changes) -> PReLU, ONE PReLU module per block used twice, ``:12-45``), BN,
  Dropout, Linear(512*8*8 -> 512), BatchNorm1d.  Same attribute names as the reference, so its ``arcface`` state dict loads.
* ``FaceLossRef`` - ``/root/reference/models/loss.py:9-78``: RGB -> gray (0.2989, 0.5870, 0.1140), bilinear resize to 128 x 128
  (``align_corners=False``), optional ``/ 127.5 - 1``, embeddings of both images, ``CosineEmbeddingLoss`` with target +1 (``maximize``)
  or -1.

Parity unpinned: the reference module cannot be imported here (its ``utils/arcface_utils.py`` needs gdown / cv2 / insightface
downloads); the building blocks are torch's own ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.PReLU`` / ``F.interpolate`` /
``nn.CosineEmbeddingLoss``, composed as the cited lines do.  Only ``tests/`` and ``__graft_entry__.smoke()`` may import this.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


class IRBlockRef(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn0 = nn.BatchNorm2d(inplanes)
        self.conv1 = _conv3x3(inplanes, inplanes)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.prelu = nn.PReLU()
        self.conv2 = _conv3x3(inplanes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):                                   # arcface_resnet.py:28-45
        out = self.prelu(self.bn1(self.conv1(self.bn0(x))))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.prelu(out + residual)


class ArcFaceResNet18Ref(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), image_size=128):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(1, 64, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.prelu = nn.PReLU()
        self.maxpool = nn.MaxPool2d(2, 2)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.bn4 = nn.BatchNorm2d(512)
        self.dropout = nn.Dropout()
        self.fc5 = nn.Linear(512 * (image_size // 16) ** 2, 512)
        self.bn5 = nn.BatchNorm1d(512)
        for m in self.modules():                              # arcface_resnet.py:85-94
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):          # arcface_resnet.py:96-109
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        layers = [IRBlockRef(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(IRBlockRef(planes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):                                     # arcface_resnet.py:111-125
        x = self.maxpool(self.prelu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.dropout(self.bn4(x))
        return self.bn5(self.fc5(x.view(x.size(0), -1)))


class FaceLossRef(nn.Module):
    def __init__(self, model=None, input_size=128):
        super().__init__()
        self.input_size = input_size
        self.model = (model if model is not None else ArcFaceResNet18Ref(image_size=input_size)).eval()
        self.cosine_loss = nn.CosineEmbeddingLoss()

    def preprocess(self, image, normalize=True):              # loss.py:26-62
        if image.size(1) == 3:
            w = torch.tensor([0.2989, 0.5870, 0.1140], device=image.device)
            image = torch.tensordot(image, w, dims=([1], [0])).unsqueeze(1)
        r = F.interpolate(image, size=(self.input_size, self.input_size), mode="bilinear", align_corners=False)
        return r / 127.5 - 1 if normalize else r

    def forward(self, x, x_gen, maximize=True, normalize=True):   # loss.py:64-78
        target = torch.ones(x.size(0)) * (1.0 if maximize else -1.0)
        return self.cosine_loss(self.model(self.preprocess(x, normalize)), self.model(self.preprocess(x_gen, normalize)), target)
