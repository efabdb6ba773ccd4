"""Optimisers over the engine's flat fp32 buffers.

TransformerOptimizer restates src/model/transformer_pytorch/optimizer.py:5-37 (Noam schedule around Adam);
FlatAdam / FlatSGD replace torch.optim.Adam(betas=(0.9, 0.98), eps=1e-9) and torch.optim.SGD(momentum,
nesterov) (call sites src/fo_meta_interface.py:105,228-236; src/transformer_torch_trainer.py:28-35) with one
streaming HIP kernel over all parameters (masr_adam_step / masr_clip_sgd_step)."""
import torch


class FlatAdam:
    def __init__(self, engine, params_flat, betas=(0.9, 0.98), eps=1e-9, lr=1e-3, weight_decay=0.0, decoupled=False):
        self.engine, self.params = engine, params_flat
        self.betas, self.eps = betas, eps
        self.weight_decay, self.decoupled = weight_decay, decoupled      # decoupled = torch.optim.AdamW, else Adam's L2 term
        self.param_groups = [{'lr': lr}]
        self.exp_avg = torch.zeros_like(params_flat)
        self.exp_avg_sq = torch.zeros_like(params_flat)
        self.t = 0
        self.grad = None                      # set by the caller before step()
        self.grad_list, self.grad_scale = None, 1.0     # ... or: several buffers whose (in-order) sum * grad_scale is the gradient
        self.inflight, self._slot = 0, 0                # guarded steps queued whose outcome (applied / skipped on NaN) is not known yet

    def zero_grad(self):
        self.grad = None
        self.grad_list = None

    def step_guarded(self, lr_applied=None, lr_skipped=None):
        """queue a step that the device skips if the gradient norm is NaN; the host learns the outcome later (`confirm`).  With one
        earlier guarded step still unconfirmed this may be step t+2 (it was applied) or t+1 (it was skipped): both sets of scalars go
        to the kernel, the earlier launch left its verdict on the device.  lr_*: the learning rate for either case (Noam)."""
        assert self.grad is not None and self.inflight <= 1, "guarded steps: at most one unconfirmed step may be in flight"
        lr = self.param_groups[0]['lr']
        self.engine.adam_step_guarded(self.params, self.grad, self.exp_avg, self.exp_avg_sq,
                                      lr if lr_applied is None else lr_applied, self.t + self.inflight + 1,
                                      lr if lr_skipped is None else lr_skipped, self.t + 1,
                                      self.betas[0], self.betas[1], self.eps, self.weight_decay, self.decoupled, self._slot)
        self._slot ^= 1
        self.inflight += 1
        if self.params.data_ptr() == self.engine.params.data_ptr():
            self.engine.mark_dirty()

    def confirm(self, applied):
        """the oldest unconfirmed guarded step was applied (its norm was a number) / skipped"""
        assert self.inflight > 0
        self.inflight -= 1
        if applied:
            self.t += 1

    def step(self):
        assert self.inflight == 0, "FlatAdam.step(): guarded steps still unconfirmed"
        if self.grad_list:
            assert not self.weight_decay, "summed-gradient step: plain Adam only"
            self.t += 1
            self.engine.adam_sum_step(self.params, self.grad_list, self.grad_scale, self.exp_avg, self.exp_avg_sq,
                                      self.param_groups[0]['lr'], self.betas[0], self.betas[1], self.eps, self.t)
            if self.params.data_ptr() == self.engine.params.data_ptr():
                self.engine.mark_dirty()
            return
        assert self.grad is not None, "FlatAdam.step(): no gradient attached"
        self.t += 1
        self.engine.adam_step(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.param_groups[0]['lr'],
                              self.betas[0], self.betas[1], self.eps, self.t, self.weight_decay, self.decoupled)
        if self.params.data_ptr() == self.engine.params.data_ptr():
            self.engine.mark_dirty()

    def state_dict(self):
        return {'t': self.t, 'exp_avg': self.exp_avg.cpu(), 'exp_avg_sq': self.exp_avg_sq.cpu(), 'lr': self.param_groups[0]['lr']}

    def load_state_dict(self, sd):
        self.t = sd['t']
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0]['lr'] = sd['lr']


class FlatRAdam(FlatAdam):
    """optimizer_cls 'RAdam' (src/transformer_torch_trainer.py:36-41): `torch_optimizer.RAdam(**optimizer_opt)` in the reference, an
    un-vendored package.  Default = that package's conventions (its authors' published implementation): rectified once N_sma >= 5,
    denom = sqrt(v) + eps (bias correction of v folded into the step size), weight decay applied to the weight (p -= lr*wd*p).
    torch_conventions=True selects torch.optim.RAdam's instead (rho_t > 5, sqrt(v)/sqrt(1-b2^t) + eps, L2 weight decay)."""

    def __init__(self, engine, params_flat, betas=(0.9, 0.999), eps=1e-8, lr=1e-3, weight_decay=0.0, torch_conventions=False):
        super().__init__(engine, params_flat, betas=betas, eps=eps, lr=lr, weight_decay=weight_decay, decoupled=not torch_conventions)
        self.variant = 0 if torch_conventions else 1

    def step(self):
        assert self.grad is not None, "FlatRAdam.step(): no gradient attached"
        self.t += 1
        self.engine.radam_step(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.param_groups[0]['lr'], self.betas[0],
                               self.betas[1], self.eps, self.t, self.weight_decay, self.variant)
        if self.params.data_ptr() == self.engine.params.data_ptr():
            self.engine.mark_dirty()


class FlatSGD:
    """torch.optim.SGD(lr, momentum, nesterov) on the engine's own params/grads; `clip` fuses clip_grad_norm_."""

    def __init__(self, engine, lr, momentum=0.0, nesterov=False, total_steps=None):
        """total_steps: the optimiser is known to be dropped after that many steps (run_task's per-task SGD): its last step does
        not write the momentum buffer, and with total_steps == 1 there is no buffer at all"""
        self.engine, self.lr, self.momentum, self.nesterov = engine, lr, momentum, nesterov
        self.param_groups = [{'lr': lr}]
        self.total_steps, self.n_steps = total_steps, 0
        self.buf = torch.zeros_like(engine.params) if momentum != 0 and total_steps != 1 else None
        self.first = True

    def zero_grad(self):
        pass                                   # run_batch overwrites every gradient

    def step(self):
        e = self.engine
        e.sgd_step(e.params, e.grads, self.buf, self.param_groups[0]['lr'], self.momentum, self.nesterov, self.first)
        self.first = False
        e.mark_dirty()

    def clip_and_step(self, max_norm):
        """clip_grad_norm_(max_norm) + `if not isnan(norm): step()` in one pass (fo_meta_interface.py:242-248)."""
        e = self.engine
        self.n_steps += 1
        flags = (1 if self.first else 0) | (2 if self.total_steps is not None and self.n_steps >= self.total_steps else 0)
        assert self.buf is not None or self.momentum == 0 or flags == 3, "FlatSGD: more steps than total_steps"
        e.clip_sgd_step(self.buf, max_norm, self.param_groups[0]['lr'], self.momentum, self.nesterov, flags)
        self.first = False

    def state_dict(self):
        return {'buf': None if self.buf is None else self.buf.cpu(), 'first': self.first, 'lr': self.param_groups[0]['lr']}

    def load_state_dict(self, sd):
        if sd['buf'] is not None:
            self.buf.copy_(sd['buf'])
        self.first, self.param_groups[0]['lr'] = sd['first'], sd['lr']


class TransformerOptimizer:
    """Noam learning-rate wrapper (optimizer.py:5-37): lr = k * d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)."""

    def __init__(self, optimizer, k, d_model, warmup_steps=25000):
        self.optimizer, self.k = optimizer, k
        self.init_lr = d_model ** (-0.5)
        self.warmup_steps = warmup_steps
        self.step_num = 0
        self.lr = self.init_lr

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        self._update_lr()
        self.optimizer.step()

    def _lr_at(self, n):
        return self.k * self.init_lr * min(n ** (-0.5), n * (self.warmup_steps ** (-1.5)))

    def _update_lr(self):
        self.step_num += 1
        self.lr = self._lr_at(self.step_num)
        for g in self.optimizer.param_groups:
            g['lr'] = self.lr

    def step_guarded(self):
        """FlatAdam.step_guarded with the schedule's learning rate for either outcome of the step still in flight"""
        self.optimizer.step_guarded(self._lr_at(self.step_num + self.optimizer.inflight + 1), self._lr_at(self.step_num + 1))

    def confirm(self, applied):
        self.optimizer.confirm(applied)
        if applied:
            self._update_lr()

    def load_state_dict(self, state_dict):
        self.optimizer.load_state_dict(state_dict)

    def state_dict(self):
        return self.optimizer.state_dict()

    def set_k(self, k):
        self.k = k
