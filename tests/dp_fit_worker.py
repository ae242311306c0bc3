"""One rank of the data-parallel `fit` check (started by test_gpu_trainer_dp.py through torch.distributed.run):
both ranks share the test box's one GPU and exchange over gloo."""
import os
import random
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V, PER_RANK, STEPS = 40, 16, 3


def shard(mols, rank, world=2):
    """Graphs of rank `rank`: global batch k holds graphs [k*world*PER_RANK, (k+1)*world*PER_RANK)."""
    G = world * PER_RANK
    return [m for k in range(len(mols) // G) for m in mols[k * G + rank * PER_RANK:k * G + (rank + 1) * PER_RANK]]


def build_model(tok):
    from ps_vae_amd import PSVAEModel, default_args, full_config
    torch.manual_seed(3)
    return PSVAEModel(full_config(default_args(), tok), tok).cuda()


def main(out_path):
    from ps_vae_amd import Tokenizer, _lib, synth
    from ps_vae_amd.data import get_dataloader
    from ps_vae_amd.parallel import GradReducer
    from ps_vae_amd.trainer import fit
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    torch.cuda.synchronize()
    _lib.check(_lib.lib().psvae_streams_init(torch.cuda.current_stream().cuda_stream), 'psvae_streams_init')
    dist.init_process_group('gloo')
    tok = Tokenizer.synthetic(V)
    train = get_dataloader(shard(synth.make_molecules(1, 'qm9', V, 2 * PER_RANK * STEPS), rank), tok, batch_size=PER_RANK,
                           shuffle=False, num_workers=0)
    valid = get_dataloader(shard(synth.make_molecules(2, 'qm9', V, 2 * PER_RANK), rank), tok, batch_size=PER_RANK,
                           shuffle=False, num_workers=0)
    model = build_model(tok)
    model.flat_buffers()             # lays the parameters out in one buffer; sets _n_trained
    random.seed(7 + rank)            # the collate's negative-pair draws
    torch.manual_seed(100 + rank)    # the latent noise
    history = fit(model, train, valid, max_epochs=1, save_dir=out_path + '.ckpt', reducer=GradReducer([0, model._n_trained]))
    torch.save({'params': model.flat_buffers()[0].detach().cpu(), 'history': history}, f'{out_path}.{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
