"""Flat parameter / gradient arenas.

All parameters of the model live in ONE fp32 buffer and all gradients in another (each
nn.Parameter is a view), laid out by `params.arena_groups`.  This is what makes the MI355X path
cheap around the kernels: one cast kernel refreshes every bf16 weight copy, gradients are zeroed
with one memset, fused QKV GEMMs read three projections as one operand, and the data-parallel
all-reduce walks a handful of contiguous buckets instead of 535 tensors.

Device-agnostic (CPU tensors work), so the multi-process gloo tests exercise the same code."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import params as P

ALIGN = 64  # elements: keeps every parameter 256-byte aligned (16-B vector loads, bf16 copy 128-B)


class FlatArena:
    def __init__(self, named_params: Dict[str, torch.nn.Parameter], groups, device=None, no_grad=()):
        """named_params: state_dict name -> Parameter (tied alias excluded).  groups: arena_groups(cfg).  no_grad: names whose
        .grad stays None as in the reference (params.no_grad_names: frozen layers, skipped connection layers)."""
        self.no_grad = frozenset(no_grad)
        self.offsets: Dict[str, Tuple[int, tuple]] = {}
        self.buckets: List[Tuple[str, int, int]] = []   # (group, lo, hi) element ranges
        off = 0
        for gname, items in groups:
            lo = off
            for name, shape in items:
                n = 1
                for s in shape:
                    n *= s
                self.offsets[name] = (off, tuple(shape))
                off += (n + ALIGN - 1) // ALIGN * ALIGN
            self.buckets.append((gname, lo, off))
        self.numel = off
        first = next(iter(named_params.values()))
        self.device = torch.device(device) if device is not None else first.device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.grad_flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.params = named_params
        with torch.no_grad():
            for name, p in named_params.items():
                o, shape = self.offsets[name]
                view = self.flat[o:o + p.numel()].view(shape)
                view.copy_(p.data.to(self.device))
                p.data = view
        self._grads_attached = False
        self.fresh = False                        # see zero_grads
        probe = []
        for _, items in groups:
            for name, _ in items:
                if name in named_params:
                    probe.append(name)
                    break
        last = [n for n in self.offsets if n in named_params][-1:]
        self._probe = list(dict.fromkeys(probe + last))

    # -- views ---------------------------------------------------------------------------------
    def view(self, name, buf=None):
        o, shape = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return (self.flat if buf is None else buf)[o:o + n].view(shape)

    def grad(self, name):
        return self.view(name, self.grad_flat)

    def is_current(self) -> bool:
        """False once something (module.to(), load of new Parameters) re-pointed the parameters."""
        base = self.flat.data_ptr()
        for name in self._probe:                      # first parameter of every bucket + the last parameter
            p = self.params[name]
            if p.device != self.flat.device or p.data_ptr() != base + 4 * self.offsets[name][0]:
                return False
        return True

    def is_current_full(self) -> bool:
        """Every parameter checked (used before an optimizer / data-parallel wrapper caches the arena)."""
        base = self.flat.data_ptr()
        return all(p.device == self.flat.device and p.data_ptr() == base + 4 * self.offsets[n][0]
                   for n, p in self.params.items())

    # -- gradients -----------------------------------------------------------------------------
    def attach_grads(self):
        """Point every used parameter's .grad at its arena slice (zeroing the arena when a grad was
        dropped, e.g. by optimizer.zero_grad(set_to_none=True))."""
        probe = None
        for name, p in self.params.items():
            if not P.is_unused(name) and name not in self.no_grad:
                probe = (name, p)
                break
        name, p = probe
        o, _ = self.offsets[name]
        ok = self._grads_attached and p.grad is not None and p.grad.data_ptr() == self.grad_flat.data_ptr() + 4 * o
        if ok:
            return
        self.grad_flat.zero_()
        self.fresh = True
        for name, p in self.params.items():
            p.grad = None if (P.is_unused(name) or name in self.no_grad) else self.grad(name)
        self._grads_attached = True

    def zero_grads(self):
        """One memset; marks the arena `fresh`: every gradient is known to be zero until the next backward has run (the engine
        then WRITES the weight gradients that have a single contributor instead of adding to them with atomics, Engine._wgrad).
        Anything else that changes gradients in place behind the arena's back (a caller adding to p.grad by hand before
        backward) must clear the flag: `arena.fresh = False`."""
        self.grad_flat.zero_()
        self.fresh = True

    def used_ranges(self) -> List[Tuple[str, int, int]]:
        return list(self.buckets)
