import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY = dict(in_channel=6, out_channel=6, inner_channel=32, norm_groups=32,
            channel_mults=(1, 2), attn_res=(8,), res_blocks=1, image_size=16)
SMALL = dict(in_channel=6, out_channel=6, inner_channel=64, norm_groups=32,
             channel_mults=(1, 2, 3, 5), attn_res=(16,), res_blocks=3, image_size=64)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
