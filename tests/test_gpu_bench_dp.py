"""The benchmark's multi-rank path on the one GPU of the test box: two ranks share the device and
exchange through gloo (RCCL needs one GPU per rank).  Checks that denominators and gradients are
reduced every step and that rank 0 prints the result line for the whole job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_two_ranks_over_gloo():
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo', PSVAE_BENCH_WATCHDOG='200')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--backend', 'gloo', '--no-cpu-baseline', '--no-bulk', '--profile-steps', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['config']['global_batch'] == 256
    assert res['value'] > 0 and res['config']['final_loss'] == res['config']['final_loss']   # finite, not NaN
    assert res['roofline']['frac'] > 0


@pytest.mark.timeout(300)
def test_bench_collectives_over_rccl_with_one_rank():
    """The RCCL code path (process group on the device, all-reduce of the flat gradient, look-ahead
    all-reduce of the denominators) with the one rank a single-GPU box allows."""
    env = dict(os.environ, PSVAE_BENCH_FORCE_DIST='1', PSVAE_BENCH_WATCHDOG='200', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29537', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-bulk',
           '--profile-steps', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert res['n_gpus'] == 1 and res['value'] > 0 and res['config']['final_loss'] == res['config']['final_loss']


def _run(env_extra, ranks, backend, port):
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo', PSVAE_BENCH_WATCHDOG='200', **env_extra)
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '6', '--warmup', '2', '--backend', backend,
            '--no-cpu-baseline', '--no-bulk', '--profile-steps', '1']
    if ranks > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
               '127.0.0.1', '--master-port', str(port)] + tail
    else:
        env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])


@pytest.mark.timeout(600)
def test_overlapped_gradient_reduce_gives_the_same_run():
    """Two gradient messages, the first started while backward is still running (on the events the step
    records), must leave the run exactly where the single message after backward leaves it."""
    plain = _run({}, 2, 'gloo', 29541)
    early = _run({'PSVAE_BENCH_OVERLAP': '1'}, 2, 'gloo', 29543)
    assert early['config']['final_loss'] == plain['config']['final_loss']
    rccl = _run({'PSVAE_BENCH_FORCE_DIST': '1', 'PSVAE_BENCH_OVERLAP': '1'}, 1, 'nccl', 29545)
    single = _run({}, 1, 'nccl', 29547)
    assert rccl['config']['final_loss'] == single['config']['final_loss']
