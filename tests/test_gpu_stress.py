"""Stress parity (GPU): many independent POA / polish groups, HIP vs the CPU oracle, bit for bit.

The small hand-made cases in test_gpu_consensus.py exercise every mode once; rare graph shapes (sibling reuse, far predecessors,
irregular rows, tile splits) only show up in volume.  This is the test that caught the topological-order violation fixed in
oracle g_add_alignment / k_poa phase A ("POA graph order").
"""
import os, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "stress_poa.py")


def _run(*args):
    p = subprocess.run([sys.executable, TOOL] + [str(a) for a in args], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(p.stdout.splitlines()[-8:])
    assert p.returncode == 0, tail


@pytest.mark.gpu
@pytest.mark.parametrize("groups,depth,seed", [(700, 8, 11), (120, 20, 5), (40, 70, 3)])
def test_spoa_groups_match_oracle(groups, depth, seed):
    _run(groups, depth, "spoa", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,depth,seed", [(80, 24, 11), (20, 90, 7)])
def test_polish_groups_match_oracle(groups, depth, seed):
    _run(groups, depth, "polish", seed)
