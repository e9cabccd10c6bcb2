"""YOLOv8n with the oriented-box (OBB) head, as a plain PyTorch module (BASELINE config 5: "PyTorch-ROCm YOLOv8n").

The reference loads its network through ultralytics (`YOLO("obb_v14.pt")`, modules/yolo.py:42-51); neither ultralytics nor the
weight file exists in its tree or in this image.  The architecture below is the published YOLOv8 definition at scale `n` (depth
0.33, width 0.25: channels 16 / 32 / 64 / 128 / 256, C2f repeats 1 / 2 / 2 / 1, SPPF, PAN neck, decoupled head with distribution
focal regression over 16 bins plus one angle channel per anchor), typed from that definition; weights are random unless a state
dict is loaded, so detections are meaningless - what config 5 exercises is the data path around the network (letterbox kernel ->
network -> decode -> rotated NMS kernel -> records -> handler).  PyTorch is plumbing here, not the product.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBNAct(nn.Module):
    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)

    def forward(self, x):
        return F.silu(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c, shortcut):
        super().__init__()
        self.cv1 = ConvBNAct(c, c, 3)
        self.cv2 = ConvBNAct(c, c, 3)
        self.add = shortcut

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    """Split, n bottlenecks on the running half, concatenate everything, 1x1."""

    def __init__(self, c1, c2, n, shortcut):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = ConvBNAct(c1, 2 * self.c, 1)
        self.cv2 = ConvBNAct((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        self.cv1 = ConvBNAct(c1, c1 // 2, 1)
        self.cv2 = ConvBNAct(c1 // 2 * 4, c2, 1)
        self.k = k

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], self.k, 1, self.k // 2))
        return self.cv2(torch.cat(y, 1))


def _branch(c_in, c_mid, c_out):
    return nn.Sequential(ConvBNAct(c_in, c_mid, 3), ConvBNAct(c_mid, c_mid, 3), nn.Conv2d(c_mid, c_out, 1))


class OBBHead(nn.Module):
    """Per level: box distribution (4 x 16 bins), class logits, one angle logit.  Output (b, 4 + nc + 1, anchors): x, y, w, h in
    input pixels, class probabilities, angle in radians in [-pi/4, 3pi/4)."""
    reg_max = 16
    strides = (8.0, 16.0, 32.0)

    def __init__(self, nc, ch):
        super().__init__()
        self.nc = nc
        c2 = max(16, ch[0] // 4, self.reg_max * 4)
        c3 = max(ch[0], min(nc, 100))
        c4 = max(ch[0] // 4, 1)
        self.box = nn.ModuleList(_branch(c, c2, 4 * self.reg_max) for c in ch)
        self.cls = nn.ModuleList(_branch(c, c3, nc) for c in ch)
        self.ang = nn.ModuleList(_branch(c, c4, 1) for c in ch)
        for b, c, s in zip(self.box, self.cls, self.strides):      # the published bias initialisation
            b[-1].bias.data[:] = 1.0
            c[-1].bias.data[:nc] = math.log(5 / nc / (640 / s) ** 2)

    def forward(self, feats):
        b = feats[0].shape[0]
        box = torch.cat([m(f).view(b, 4 * self.reg_max, -1) for m, f in zip(self.box, feats)], 2)
        cls = torch.cat([m(f).view(b, self.nc, -1) for m, f in zip(self.cls, feats)], 2)
        ang = torch.cat([m(f).view(b, 1, -1) for m, f in zip(self.ang, feats)], 2)
        angle = (ang.sigmoid() - 0.25) * math.pi
        anchors, stride = [], []
        for f, s in zip(feats, self.strides):
            h, w = f.shape[2:]
            sy, sx = torch.meshgrid(torch.arange(h, device=f.device, dtype=f.dtype) + 0.5, torch.arange(w, device=f.device, dtype=f.dtype) + 0.5,
                                    indexing="ij")
            anchors.append(torch.stack((sx, sy), -1).view(-1, 2))
            stride.append(torch.full((h * w, 1), s, device=f.device, dtype=f.dtype))
        anchors = torch.cat(anchors).t().unsqueeze(0)               # (1, 2, A)
        stride = torch.cat(stride).t().unsqueeze(0)                 # (1, 1, A)
        # distribution focal regression: expectation over the 16 bins of each side
        dist = box.view(b, 4, self.reg_max, -1).softmax(2)
        dist = (dist * torch.arange(self.reg_max, device=box.device, dtype=box.dtype).view(1, 1, -1, 1)).sum(2)   # (b, 4, A): l, t, r, b
        lt, rb = dist[:, :2], dist[:, 2:]
        cos, sin = torch.cos(angle), torch.sin(angle)
        xf, yf = ((rb - lt) / 2).split(1, 1)
        xy = torch.cat([xf * cos - yf * sin, xf * sin + yf * cos], 1) + anchors
        xywh = torch.cat([xy, lt + rb], 1) * stride
        return torch.cat([xywh, cls.sigmoid(), angle], 1)


class YOLOv8nOBB(nn.Module):
    def __init__(self, nc):
        super().__init__()
        c = (16, 32, 64, 128, 256)
        self.b0 = ConvBNAct(3, c[0], 3, 2)
        self.b1 = ConvBNAct(c[0], c[1], 3, 2)
        self.b2 = C2f(c[1], c[1], 1, True)
        self.b3 = ConvBNAct(c[1], c[2], 3, 2)
        self.b4 = C2f(c[2], c[2], 2, True)
        self.b5 = ConvBNAct(c[2], c[3], 3, 2)
        self.b6 = C2f(c[3], c[3], 2, True)
        self.b7 = ConvBNAct(c[3], c[4], 3, 2)
        self.b8 = C2f(c[4], c[4], 1, True)
        self.b9 = SPPF(c[4], c[4], 5)
        self.h12 = C2f(c[4] + c[3], c[3], 1, False)
        self.h15 = C2f(c[3] + c[2], c[2], 1, False)
        self.h16 = ConvBNAct(c[2], c[2], 3, 2)
        self.h18 = C2f(c[2] + c[3], c[3], 1, False)
        self.h19 = ConvBNAct(c[3], c[3], 3, 2)
        self.h21 = C2f(c[3] + c[4], c[4], 1, False)
        self.head = OBBHead(nc, (c[2], c[3], c[4]))

    def forward(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        n4 = self.h12(torch.cat([F.interpolate(p5, scale_factor=2.0, mode="nearest"), p4], 1))
        n3 = self.h15(torch.cat([F.interpolate(n4, scale_factor=2.0, mode="nearest"), p3], 1))
        o4 = self.h18(torch.cat([self.h16(n3), n4], 1))
        o5 = self.h21(torch.cat([self.h19(o4), p5], 1))
        return self.head([n3, o4, o5])
