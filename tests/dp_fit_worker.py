"""One rank of the data-parallel `fit` check (started by test_gpu_trainer_dp.py through torch.distributed.run):
both ranks share the test box's one GPU and exchange over gloo."""
import os
import random
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V, PER_RANK, STEPS, EPOCHS = 40, 16, 3, 2


def shard(mols, rank, world=2):
    """Graphs of rank `rank`: global batch k holds graphs [k*world*PER_RANK, (k+1)*world*PER_RANK)."""
    G = world * PER_RANK
    return [m for k in range(len(mols) // G) for m in mols[k * G + rank * PER_RANK:k * G + (rank + 1) * PER_RANK]]


def build_model(tok):
    from ps_vae_amd import PSVAEModel, default_args, full_config
    torch.manual_seed(3)
    return PSVAEModel(full_config(default_args(), tok), tok).cuda()


def resident_batches(tok, rank, device):
    """This rank's training and validation batches, collated once and kept on the device: the same batch objects
    come round again every epoch (as a benchmark or a small data set keeps them)."""
    from ps_vae_amd import synth
    from ps_vae_amd.data import get_dataloader, to_device
    random.seed(7 + rank)            # the collate's negative-pair draws
    train = get_dataloader(shard(synth.make_molecules(1, 'qm9', V, 2 * PER_RANK * STEPS), rank), tok, batch_size=PER_RANK,
                           shuffle=False, num_workers=0)
    valid = get_dataloader(shard(synth.make_molecules(2, 'qm9', V, 2 * PER_RANK), rank), tok, batch_size=PER_RANK,
                           shuffle=False, num_workers=0)
    return [to_device(b, device) for b in train], [to_device(b, device) for b in valid]


def main(out_path):
    from ps_vae_amd import Tokenizer, _lib
    from ps_vae_amd.parallel import OverlappedReducer, make_reducer
    from ps_vae_amd.trainer import fit
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    torch.cuda.synchronize()
    _lib.check(_lib.lib().psvae_streams_init(torch.cuda.current_stream().cuda_stream), 'psvae_streams_init')
    dist.init_process_group('gloo')
    tok = Tokenizer.synthetic(V)
    train, valid = resident_batches(tok, rank, 'cuda')
    model = build_model(tok)
    reducer = make_reducer(model)    # the default: three gradient messages, two of them overlapped with backward
    assert isinstance(reducer, OverlappedReducer)
    history = fit(model, train, valid, max_epochs=EPOCHS, save_dir=out_path + '.ckpt', reducer=reducer)
    draws = model.eps_draws
    z = model.get_z(train[0])        # a sampled latent: draws the epsilon of this rank's graphs of the global batch
    mean, lvar = model.encode(train[0])
    torch.save({'params': model.flat_buffers()[0].detach().cpu(), 'history': history, 'z': z.cpu(), 'draws': draws,
                'mean': mean.cpu(), 'log_var': lvar.cpu(), 'graph_base': train[0]['_native'].graph_base,
                'denoms': train[0]['_native'].denoms.cpu(), 'local_denoms': train[0]['_native'].local_denoms.cpu()},
               f'{out_path}.{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
