"""Fused global-norm clip + Adam over the model's flat buffers: the semantics of
`clip_grad_norm_(params, 10.0)` (`src/train.py:47`) followed by
`torch.optim.Adam(lr)` with the per-epoch `lr / (epoch + 1)` schedule
(`src/pl_models/ps_vae_model.py:123-127`), as two kernel launches and no host
synchronisation."""
import torch

from . import _lib


class FusedClipAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=10.0):
        self.model, self.base_lr, self.betas, self.eps, self.max_norm = model, lr, betas, eps, max_norm
        self.epoch = 0
        self.step_count = 0
        self.exp_avg = self.exp_avg_sq = self.partials = self.grad_norm = None

    @property
    def lr(self):
        return self.base_lr / (self.epoch + 1)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def _state(self, params):
        if self.exp_avg is None or self.exp_avg.device != params.device or self.exp_avg.numel() != params.numel():
            self.exp_avg = torch.zeros_like(params)
            self.exp_avg_sq = torch.zeros_like(params)
        if self.partials is None or self.partials.device != params.device:      # also after load_state_dict
            L = _lib.lib()
            self.partials = torch.zeros(L.psvae_adam_partials(params.numel()), dtype=torch.float32, device=params.device)
            self.grad_norm = torch.zeros(1, dtype=torch.float32, device=params.device)

    def step(self):
        L = _lib.lib()
        params, grads = self.model.flat_buffers()
        if not params.is_cuda:
            raise RuntimeError('FusedClipAdam runs on the GPU only')
        self._state(params)
        self.step_count += 1
        b1, b2 = self.betas
        bias1, bias2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        _lib.check(L.psvae_clip_adam(params.data_ptr(), grads.data_ptr(), self.exp_avg.data_ptr(),
                                     self.exp_avg_sq.data_ptr(), params.numel(), self.partials.data_ptr(),
                                     self.max_norm, self.lr, b1, b2, self.eps, bias1, bias2, self.grad_norm.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), 'psvae_clip_adam')
        self.model.zero_grad()

    def fused_struct(self):
        """The `psvae_adam` block of the NEXT step (advances the step count): what `PSVAEModel.train_step` hands to
        `psvae_train_step` so that clip + Adam are queued by the same library call as forward and backward."""
        params, grads = self.model.flat_buffers()
        if not params.is_cuda:
            raise RuntimeError('FusedClipAdam runs on the GPU only')
        self._state(params)
        self.step_count += 1
        b1, b2 = self.betas
        a = _lib.Adam()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = params.data_ptr(), grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.count, a.partials = params.numel(), self.partials.data_ptr()
        a.max_norm, a.lr, a.beta1, a.beta2, a.eps = self.max_norm, self.lr, b1, b2, self.eps
        a.bias1, a.bias2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        a.grad_norm_out = self.grad_norm.data_ptr()
        return a

    def state_dict(self):
        """Step count, epoch and Adam's two moments PER PARAMETER NAME (views of the flat buffers): a checkpoint then
        does not depend on the order in which this version lays the parameters out in its flat buffer."""
        out = {'step': self.step_count, 'epoch': self.epoch, 'exp_avg': None, 'exp_avg_sq': None}
        if self.exp_avg is not None:
            model = self.model
            model.flat_buffers()
            named = dict(model.named_parameters())
            for key, buf in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                out[key] = {k: buf[model._offsets[k]:model._offsets[k] + named[k].numel()].view(named[k].shape)
                            for k in model.trained_names()}
        return out

    def load_state_dict(self, sd):
        """Accepts `state_dict()`'s per-name moments; a flat moment buffer (what versions before the per-name form
        wrote) only together with `flat_order`, the list of parameter names in the order that buffer was laid out -
        without it the buffer cannot be told apart from one in another order of the same length, and is refused."""
        self.step_count, self.epoch = int(sd['step']), int(sd['epoch'])
        if sd.get('exp_avg') is None:
            self.exp_avg = self.exp_avg_sq = None
            return
        model = self.model
        params, _ = model.flat_buffers()
        named = dict(model.named_parameters())
        new = {}
        for key in ('exp_avg', 'exp_avg_sq'):
            src = sd[key]
            flat = torch.zeros_like(params)
            if torch.is_tensor(src):
                order = sd.get('flat_order')
                if order is None:
                    raise ValueError('optimizer state holds a flat moment buffer without its parameter order '
                                     "('flat_order'): it cannot be mapped onto this version's layout")
                if sorted(order) != sorted(model.trained_names()):
                    raise ValueError("'flat_order' names other parameters than this model trains")
                off, pieces = 0, {}
                for k in order:
                    pieces[k] = src[off:off + named[k].numel()]
                    off += (named[k].numel() + 3) // 4 * 4
                src = pieces
            missing = [k for k in model.trained_names() if k not in src]
            if missing:
                raise ValueError(f'optimizer state lacks moments for {missing[:3]}...')
            for k in model.trained_names():
                o = model._offsets[k]
                flat[o:o + named[k].numel()] = src[k].reshape(-1).to(params.device, torch.float32)
            new[key] = flat
        self.exp_avg, self.exp_avg_sq = new['exp_avg'], new['exp_avg_sq']
        self.partials = None
