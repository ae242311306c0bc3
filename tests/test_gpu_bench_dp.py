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
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it and no WORLD_SIZE in the environment: the program
    itself starts the two rank processes; its standard output is rank 0's single result line, for the whole job."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GLOO_SOCKET_IFNAME='lo')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--backend', 'gloo',
           '--watchdog', '200', '--no-configs', '--no-cpu-baseline', '--no-bulk', '--profile-steps', '1', '--config3-steps', '3']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['config']['global_batch'] == 256
    assert res['rccl_ranks'] == 0                        # gloo rehearsal: no RCCL rank took part
    assert res['value'] > 0 and res['config']['final_loss'] == res['config']['final_loss']   # finite, not NaN
    assert res['roofline']['frac'] > 0
    strong = res['strong_scaling']                       # vocab 500, global batch 1024 -> 512 graphs per rank
    assert strong['scaling'] == 'strong' and '512 graphs per GPU' in strong['workload'] and strong['value'] > 0


@pytest.mark.timeout(300)
def test_bench_two_ranks_under_a_launcher():
    """The driver's form: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE."""
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--backend', 'gloo', '--watchdog', '200', '--no-configs', '--no-cpu-baseline', '--no-bulk', '--profile-steps', '1', '--scaling', 'strong',
           '--global-batch', '96', '--vocab', '500', '--config3-steps', '0']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'strong' and res['config']['global_batch'] == 96
    assert 'batch 48 per GPU' in res['config']['workload'] and res['value'] > 0


@pytest.mark.timeout(300)
def test_bench_collectives_over_rccl_with_one_rank():
    """The RCCL code path (process group on the device, all-reduce of the flat gradient, look-ahead
    all-reduce of the denominators) with the one rank a single-GPU box allows."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29537', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-bulk',
           '--no-configs', '--force-dist', '--watchdog', '200', '--profile-steps', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert res['n_gpus'] == 1 and res['value'] > 0 and res['config']['final_loss'] == res['config']['final_loss']


def _run(flags, ranks, backend, port, extra=()):
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo')
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '6', '--warmup', '2', '--backend', backend,
            '--watchdog', '200', '--no-configs', '--no-cpu-baseline', '--no-bulk', '--profile-steps', '1', '--config3-steps', '0'] + list(flags) + list(extra)
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
    """Three gradient messages, the first two started while backward is still running (on the events the step
    records: the default), must leave the run exactly where the single message after backward leaves it; and two
    ranks on the halves of a batch must end where one process ends on the whole batch (same graphs, same
    negative pairs, same epsilon per graph), up to the order of fp32 sums."""
    early = _run([], 2, 'gloo', 29543)
    plain = _run(['--dp-overlap', '0'], 2, 'gloo', 29541)
    assert 'OverlappedReducer' in early['config']['grad_reduce'] and 'GradReducer' in plain['config']['grad_reduce']
    assert early['config']['final_loss'] == plain['config']['final_loss']
    once = _run(['--denoms', 'once'], 2, 'gloo', 29549)
    assert once['config']['final_loss'] == plain['config']['final_loss']
    rccl = _run(['--force-dist'], 1, 'nccl', 29545)
    single = _run([], 1, 'nccl', 29547)
    assert rccl['rccl_ranks'] == 1 and single['rccl_ranks'] == 0
    assert rccl['config']['final_loss'] == single['config']['final_loss']
    whole = _run([], 1, 'nccl', 29551, extra=['--batch-size', '256'])
    assert abs(early['config']['final_loss'] - whole['config']['final_loss']) <= 2e-4 * abs(whole['config']['final_loss'])
