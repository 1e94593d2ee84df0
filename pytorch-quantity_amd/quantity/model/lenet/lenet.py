"""LeNet-style MNIST classifier in "fabu" style (View marker module for the flatten): the small
Conv + Linear fixture -- biased convolutions, max-pooling between cared layers, three chained Linear
layers.

Module tree as in the reference's quantity/model/lenet/lenet.py (Cnn :12-30): `conv` is a Sequential
(Conv2d, ReLU, MaxPool2d) x 2 so the cared layers are conv.0 and conv.3; `review` is the View;
`fc` is a Sequential of Linear(400,120), Linear(120,84), Linear(84,n_class).
"""
import sys

from torch import nn

sys.path.insert(0, '../../')
from common.quantity import View  # noqa: E402

_FEATURES = ((6, 3, 1), (16, 5, 0))      # (out channels, kernel, padding) of the two conv stages
_HIDDEN = (400, 120, 84)


class Cnn(nn.Module):

    def __init__(self, in_dim, n_class):
        super(Cnn, self).__init__()
        layers, cin = [], in_dim
        for cout, k, pad in _FEATURES:
            layers += [nn.Conv2d(cin, cout, k, stride=1, padding=pad), nn.ReLU(False), nn.MaxPool2d(2, 2)]
            cin = cout
        self.conv = nn.Sequential(*layers)
        self.review = View()
        widths = _HIDDEN + (n_class,)
        self.fc = nn.Sequential(*[nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])])

    def forward(self, x):
        flat = self.review(self.conv(x))
        return self.fc(flat)
