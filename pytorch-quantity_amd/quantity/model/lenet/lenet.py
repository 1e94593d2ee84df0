"""LeNet-style MNIST classifier written with the View marker layer: the small Conv + Linear fixture
(biased convolutions, max-pooling between cared layers, a chain of Linear layers).

Same module tree as the reference's quantity/model/lenet/lenet.py (Cnn :12-30): conv (Sequential of
Conv2d/ReLU/MaxPool2d x2), review (View), fc (Sequential of three Linear)."""
import sys

from torch import nn

sys.path.insert(0, '../../')
from common.quantity import View  # noqa: E402


class Cnn(nn.Module):

    def __init__(self, in_dim, n_class):
        super(Cnn, self).__init__()
        features = [nn.Conv2d(in_dim, 6, 3, stride=1, padding=1), nn.ReLU(False), nn.MaxPool2d(2, 2),
                    nn.Conv2d(6, 16, 5, stride=1, padding=0), nn.ReLU(False), nn.MaxPool2d(2, 2)]
        self.conv = nn.Sequential(*features)
        self.review = View()
        self.fc = nn.Sequential(nn.Linear(400, 120), nn.Linear(120, 84), nn.Linear(84, n_class))

    def forward(self, x):
        return self.fc(self.review(self.conv(x)))
