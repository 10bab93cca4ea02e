"""Cross-entropy for the executor's logits in two small launches (torch.nn.CrossEntropyLoss of
Train_model.py:185,312 takes seven: log-softmax, gather, mean, and their gradients).

    loss_fn = vqa_amd.loss.CrossEntropyLoss()
    loss_fn(scores, answers).backward()

Mean reduction over the batch, no class weights, no label smoothing, targets int64 in [0, num_answers):
the reference's call.  The forward kernel also writes d(loss)/d(logits); backward multiplies it by the
incoming gradient.  fp32 logits on the GPU only: there is no CPU path.
"""
import torch

from . import ops


class _Xent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        if not logits.is_cuda or logits.dim() != 2 or logits.dtype != torch.float32:
            raise ValueError('logits must be a 2-D fp32 tensor on the GPU')
        if target.dim() != 1 or target.size(0) != logits.size(0):
            raise ValueError('targets must be [N]')
        lg = logits.detach().contiguous()
        tg = target.detach().to(device=logits.device, dtype=torch.int64).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        d = torch.empty_like(lg)
        ops.xent(lg, tg, loss, d)
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, grad):
        (d,) = ctx.saved_tensors
        return d * grad, None


def cross_entropy(logits, target):
    return _Xent.apply(logits, target)


class CrossEntropyLoss(torch.nn.Module):
    def forward(self, logits, target):
        return _Xent.apply(logits, target)
