"""ps_vae_amd: the PS-VAE encoder/decoder training step as hand-written gfx950
(MI355X) HIP kernels behind the reference's nn.Module surface."""
from .tokenizer import Tokenizer, GeneralVocab  # noqa: F401
from .nn_config import (common_config, predictor_config, encoder_config, ps_vae_config, full_config,  # noqa: F401
                        default_args)


def __getattr__(name):
    # torch-dependent parts are imported lazily so that `import ps_vae_amd` stays cheap
    if name in ('PSVAEModel',):
        from .model import PSVAEModel
        return PSVAEModel
    if name in ('Encoder', 'VAEPieceDecoder', 'Predictor', 'MLP'):
        from . import modules
        return getattr(modules, name)
    if name in ('BPEMolDataset', 'PackedMolDataset', 'get_dataloader', 'to_device', 'NativeBatch'):
        from . import data
        return getattr(data, name)
    if name == 'DeviceMolDataset':
        from .device_data import DeviceMolDataset
        return DeviceMolDataset
    if name == 'FusedClipAdam':
        from .optim import FusedClipAdam
        return FusedClipAdam
    raise AttributeError(name)
