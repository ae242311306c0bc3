"""The C ABI without a GPU: libpsvae_hip.so loads, exports every function include/psvae.h declares,
the ctypes mirrors of its structs have the header's field counts, argument errors are reported by
code before anything is launched, and the product path refuses to run without a GPU instead of
falling back to anything else."""
import ctypes as C
import os
import re

import pytest
import torch

from ps_vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, 'include', 'psvae.h')).read()


def declared_functions():
    body = re.sub(r'/\*.*?\*/', '', HEADER, flags=re.S)
    return sorted(set(re.findall(r'\b(psvae_[a-z0-9_]+)\s*\(', body)))


def struct_fields(name):
    body = re.sub(r'/\*.*?\*/', '', HEADER, flags=re.S)
    block = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), body, flags=re.S).group(1)
    n = 0
    for decl in block.split(';'):
        decl = decl.strip()
        if decl:
            n += decl.count(',') + 1
    return n


def test_library_exports_every_declared_function():
    L = _lib.lib()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names, set(names) ^ set(_lib.EXPORTS)
    assert L.psvae_abi_version() == int(re.search(r'#define PSVAE_ABI_VERSION (\d+)', HEADER).group(1))


def test_ctypes_structs_mirror_the_header():
    assert len(_lib.Dims._fields_) == struct_fields('psvae_dims')
    assert len(_lib.Params._fields_) == struct_fields('psvae_params')
    assert len(_lib.Batch._fields_) == struct_fields('psvae_batch')
    assert len(_lib.MolStore._fields_) == struct_fields('psvae_mol_store')
    assert C.sizeof(_lib.Params) == 8 * len(_lib.Params._fields_)
    assert len(_lib.BLOB_FIELDS) == len(re.findall(r'PSVAE_BLOB_[A-Z_]+', HEADER.split('enum {')[-1].split('}')[0])) - 1


def test_argument_errors_come_back_as_codes():
    L = _lib.lib()
    assert L.psvae_strerror(0) and L.psvae_strerror(-1) != L.psvae_strerror(-2)
    assert L.psvae_linear_forward(None, 4, None, None, None, 4, 8, 4, 4, 0, None, 0, None) == -1          # PSVAE_E_NULL
    assert L.psvae_gru_forward(None, None, None, None, None, None, None, 4, 2, 8, None, 0, None) == -1
    dims, batch = _lib.Dims(), _lib.Batch()
    assert L.psvae_workspace_bytes(C.byref(dims), C.byref(batch)) == 0                                    # rejected sizes
    assert L.psvae_step_forward(C.byref(dims), None, C.byref(batch), None, 0, None, None, None) < 0
    with pytest.raises(RuntimeError):
        _lib.check(-2, 'probe')


def test_options_table():
    """Kernel-selection options: every name the header documents exists, values can be set and read back inside one
    process, -1 (and anything below) means "the library chooses", unknown names are an error code."""
    L = _lib.lib()
    names = _lib.option_names()
    documented = re.findall(r'^ \*   ([a-z_0-9]+) +(?:-?\d|bound)', HEADER, flags=re.M)
    assert sorted(names) == sorted(documented) and 'split_bf16' in names and 'gru_spin_ticks' in names
    for n in names:
        env = os.environ.get('PSVAE_' + n.upper())
        old = _lib.set_option(n, 3)
        assert old == (int(env) if env else -1)
        assert _lib.set_option(n, -7) == 3
        assert _lib.set_option(n, old) == -1
    v = C.c_int(0)
    assert L.psvae_option_set(b'no_such_option', 1) == -2 and L.psvae_option_get(b'no_such_option', C.byref(v)) == -2
    assert L.psvae_option_get(b'streams', None) == -1


def test_no_cpu_path():
    """Without a GPU the model raises: there is no CPU implementation behind the module surface."""
    from ps_vae_amd import PSVAEModel, Tokenizer, default_args, full_config
    tok = Tokenizer.synthetic(30)
    model = PSVAEModel(full_config(default_args(), tok), tok)
    assert len(model.state_dict()) == 51
    with pytest.raises(RuntimeError, match='GPU'):
        model({'x': torch.zeros(1, 1, dtype=torch.long)})


def test_device_collate_layout_is_the_host_builders_layout():
    """The blob layout the GPU batch builder is told to fill (a host-side function) equals the layout
    psvae_pack_batch produces for the same counts."""
    import random
    from ps_vae_amd import Tokenizer, synth
    from ps_vae_amd.data import BPEMolDataset, PackedMolDataset
    tok = Tokenizer.synthetic(300)
    random.seed(3)
    nb = PackedMolDataset(synth.make_molecules(9, 'zinc', 300, 24), tok).collate_fn(list(range(24)))['_native']
    sz = nb.sizes
    off = (C.c_int64 * len(_lib.BLOB_FIELDS))()
    words = _lib.lib().psvae_collate_device_layout(sz['B'], sz['n'], sz['E'], sz['E_in'], sz['S'], sz['L'], tok.num_atom_type(),
                                                   tok.num_piece_type(), BPEMolDataset.max_pos, off)
    assert list(off) == list(nb.offsets) and words == nb.blob.numel()
    assert _lib.lib().psvae_collate_device_layout(0, 1, 0, 0, 0, 2, 11, 302, 50, off) == -1
    assert _lib.lib().psvae_collate_device(None, None, 1, 1, 0, 0, 2, 1, 0.0, 0, 0, 11, 302, 50, 300, None, 1, None, 0, None,
                                          None, None, None, 0, None) == -1          # PSVAE_E_NULL before any launch
