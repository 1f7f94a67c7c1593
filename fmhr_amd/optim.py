"""fmhr_amd.optim — Adam with the semantics of `torch.optim.Adam` as the HAM loop uses it (/root/reference/mesh_sfs_optim.py:193,
244: per-group lr, default betas/eps, no weight decay, no amsgrad), one small HIP kernel per parameter instead of torch's
multi-tensor launch (≈94 us per group at V = 49k; the whole stage-2 step is ≈1.5 ms)."""
import torch

from ._lib import call, ptr, stream_ptr


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{'params': groups}]
        self.param_groups = []
        for g in groups:
            ps = g['params']
            ps = [ps] if isinstance(ps, torch.Tensor) else list(ps)
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("fmhr_amd.optim.Adam: parameters must be contiguous float32 GPU tensors")
            self.param_groups.append(dict(params=ps, lr=float(g.get('lr', lr)), betas=tuple(g.get('betas', betas)),
                                          eps=float(g.get('eps', eps))))
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    def fused_param(self, p):
        """The C-ABI form of ONE pending update of parameter `p` (include/fmhr_ham.h: fmhr_adam_param) for the kernels that apply
        Adam where the gradient is formed (HamStage2.step(optimizer=...)): advances p's step count exactly as step() would."""
        for g in self.param_groups:
            if any(q is p for q in g['params']):
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                st['step'] += 1
                from ._lib import AdamParam
                return AdamParam(p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), g['lr'], g['betas'][0],
                                 g['betas'][1], g['eps'], st['step'])
        raise RuntimeError("fmhr_amd.optim.Adam.fused_param: the tensor is not a parameter of this optimiser")

    @torch.no_grad()
    def step(self):
        todo = []
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None or not p.requires_grad:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                st['step'] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                todo.append((p, grad, st, g))

        def args(t):
            p, grad, st, g = t
            return (ptr(p), ptr(grad), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), p.numel(), g['lr'], g['betas'][0], g['betas'][1],
                    g['eps'], st['step'])

        i = 0
        while i + 1 < len(todo) and todo[i][0].device == todo[i + 1][0].device:      # pairs share one launch
            call("fmhr_adam_step2", *args(todo[i]), *args(todo[i + 1]), stream_ptr(todo[i][0].device))
            i += 2
        for t in todo[i:]:
            call("fmhr_adam_step", *args(t), stream_ptr(t[0].device))
