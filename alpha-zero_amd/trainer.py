"""Training step of the reference (train.train, train.py:85-123) for the functional ViT, data-parallel over GPUs.

Same loss as the reference, term by term:
    policy_loss = mean_b( sum_a( -pi[b,a] * log_softmax(logits)[b,a] ) )
    value_loss  = mse(values[B,1], rewards[B,1])
    l2          = sum of squares of every parameter whose NAME does not contain 'bias' or 'LayerNorm'
                  (the modules are called norm1 / norm2 / norm, so LayerNorm WEIGHTS are regularised - SURVEY 3.3)
    loss        = policy_loss + value_loss + 1e-4 * l2
and the same optimiser life cycle: a fresh Adam(lr) for every train() call (train.py:93).

Data parallel: every rank computes the loss on its shard of the batch; gradients are flattened into ONE bucket
(3.41 M fp32 = 13.65 MB at the training config), all-reduced (RCCL over xGMI on GPUs, gloo in the CPU test), divided
by the world size and scattered back - no per-tensor collectives, no overlap machinery needed at this size.
"""
import torch
import torch.nn.functional as F

from pvnet import PolicyValueNet, reference_key_shapes


class Trainer:
    def __init__(self, cfg, weights, device="cpu", dropout=None):
        # dropout: Net(..., dropout=) of the reference (main.py:134 trains with 0.1 under model.train(), train.py:92);
        # default = the configuration's own value
        self.cfg, self.device = cfg, torch.device(device)
        self.dropout = float(getattr(cfg, "dropout", 0.0) if dropout is None else dropout)
        order = list(reference_key_shapes(cfg))                       # == named_parameters() order of the reference Net
        self.params = {k: torch.nn.Parameter(torch.as_tensor(weights[k]).detach().to(self.device, torch.float32).clone())
                       for k in order}
        self.net = PolicyValueNet(cfg, weights={k: v.detach().cpu() for k, v in self.params.items()}, device="cpu",
                                  dtype=torch.float32, path="full")
        self.net.device, self.net.w = self.device, self.params        # the forward reads these tensors directly

    def state_dict(self):
        return {k: v.detach().cpu().clone() for k, v in self.params.items()}

    def loss_terms(self, states, pis, rewards):
        logits, values = self.net.forward_impl(states.to(self.device, torch.float32), "full", dropout_p=self.dropout)
        l2 = 0.0
        for name, p in self.params.items():                           # train.py:101-108
            if "bias" in name or "LayerNorm" in name:
                continue
            l2 = l2 + torch.sum(p ** 2)
        log_probs = F.log_softmax(logits, dim=1)
        policy_loss = torch.mean(torch.sum(-pis.to(self.device, torch.float32) * log_probs, dim=1))
        value_loss = F.mse_loss(values, rewards.to(self.device, torch.float32))
        return policy_loss + value_loss + 1e-4 * l2, policy_loss, value_loss, l2

    def allreduce_grads(self, dist):
        """One fused bucket: flatten -> all_reduce(SUM) -> / world -> scatter back."""
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return
        grads = [p.grad for p in self.params.values()]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def train(self, batches, lr, dist=None):
        """batches: iterable of (states [B,F,R,C], pis [B,A], rewards [B,1]) - this rank's shard of each iteration.
        Returns the last iteration's (loss, policy_loss, value_loss, l2) like train.train does."""
        opt = torch.optim.Adam(list(self.params.values()), lr=lr)     # fresh optimiser per call (train.py:93)
        last = None
        for states, pis, rewards in batches:
            loss, pl, vl, l2 = self.loss_terms(states, pis, rewards)
            opt.zero_grad()
            loss.backward()
            self.allreduce_grads(dist)
            opt.step()
            last = (loss.item(), pl.item(), vl.item(), float(l2.detach()))
        return last
