import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def run_in_fresh_process(test_file, test_name, timeout=900):
    """Tests that CAPTURE hipGraphs run in a process of their own (round 6): a graph captured after the nested stage fork (tcct_amd.ops.STAGE_FORK_MAX_PIXELS, on by
    default, used by nearly every earlier test of a suite run) crashes in hipGraphLaunch, and tcct_amd.graph refuses such a capture.  The parent test calls this
    and returns when it gives True; the child (TCCT_TEST_CHILD=1) runs the test body.  One child at a time; the child is this same pytest invocation of ONE test."""
    import subprocess
    if os.environ.get('TCCT_TEST_CHILD') == '1':
        return False
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', f'{test_file}::{test_name}'],
                       env=dict(os.environ, TCCT_TEST_CHILD='1'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, f'{test_name} in a fresh process: rc {r.returncode}\n' + r.stdout[-4000:]
    return True
