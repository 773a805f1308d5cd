"""Synthetic model definitions for the BASELINE.json configs (random init, no datasets).

ResNet-18 follows the torchvision topology (torchvision itself is not installed here), with the
CIFAR-style `num_classes=10` head the reference's benchmark uses
(`docs/examples/basic_usage/benchmark_utils.py:341-452`)."""

from __future__ import annotations

import torch
from torch import nn


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNet18(nn.Module):
    def __init__(self, num_classes: int = 10):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cfg = [(64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1),
               (256, 512, 2), (512, 512, 1)]
        self.layers = nn.Sequential(*[BasicBlock(a, b, s) for a, b, s in cfg])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layers(x)
        return self.fc(torch.flatten(self.avgpool(x), 1))


class ResNetToy(nn.Module):
    """The ResNet-18 building blocks at toy size (golden-vector fixture `tests/golden/nets.npz`): stem
    conv + BatchNorm, a plain BasicBlock, a strided BasicBlock with a 1x1 down-sampling branch, global
    average pool, Linear head."""

    def __init__(self, classes: int = 5):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 4, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(4)
        self.relu = nn.ReLU()
        self.layers = nn.Sequential(BasicBlock(4, 4, 1), BasicBlock(4, 6, 2))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(6, classes)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        return self.fc(torch.flatten(self.avgpool(self.layers(x)), 1))


def lenet5() -> nn.Sequential:
    return nn.Sequential(
        nn.Conv2d(1, 6, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2),
        nn.Flatten(), nn.Linear(400, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, 10),
    )


def kfac_params(model: nn.Module) -> dict[str, torch.Tensor]:
    """Linear / Conv2d parameters only (the reference's KFAC benchmarks exclude BatchNorm,
    `benchmark_execute.py:172-183`)."""
    keep = {}
    for mod_name, mod in model.named_modules():
        if isinstance(mod, (nn.Linear, nn.Conv2d)):
            for p_name, p in mod.named_parameters(recurse=False):
                keep[f"{mod_name}.{p_name}" if mod_name else p_name] = p
    return keep


class EncoderBlock(nn.Module):
    """Pre-LN transformer block with explicit (math) attention, so that forward- and reverse-mode
    `torch.func` transforms apply."""

    def __init__(self, d: int, heads: int, ffn: int):
        super().__init__()
        self.h = heads
        self.ln1, self.ln2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, ffn), nn.Linear(ffn, d)

    def forward(self, x):
        B, S, d = x.shape
        q, k, v = self.qkv(self.ln1(x)).view(B, S, 3, self.h, d // self.h).permute(2, 0, 3, 1, 4)
        att = torch.softmax(q @ k.transpose(-1, -2) / (d // self.h) ** 0.5, dim=-1)
        x = x + self.proj((att @ v).transpose(1, 2).reshape(B, S, d))
        return x + self.fc2(torch.nn.functional.gelu(self.fc1(self.ln2(x))))


class Encoder(nn.Module):
    """BASELINE config C5: 12 pre-LN blocks, d = 768, 12 heads, ffn 3072, mean-pool + Linear(768, 10)."""

    def __init__(self, d: int = 768, layers: int = 12, heads: int = 12, ffn: int = 3072, classes: int = 10):
        super().__init__()
        self.blocks = nn.Sequential(*[EncoderBlock(d, heads, ffn) for _ in range(layers)])
        self.ln = nn.LayerNorm(d)
        self.head = nn.Linear(d, classes)

    def forward(self, x):
        return self.head(self.ln(self.blocks(x)).mean(dim=1))


def encoder_toy() -> Encoder:
    """The C5 encoder at toy size (golden-vector fixture `tests/golden/nets.npz`)."""
    return Encoder(d=32, layers=2, heads=4, ffn=64, classes=5)
