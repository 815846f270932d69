"""Flat fp32 parameter arena in HBM.

All trainable tensors live in ONE contiguous device buffer (each tensor padded to a multiple of 4 floats so every view
is 16-byte aligned), in TF variable-creation order (spec.param_table).  A second buffer of identical layout holds the
gradients, a third the BN moving statistics.  `VariableState.export_variables/import_variables`
(meta_learners/variables.py:58-80) become device-side clones/copies of these buffers, the Reptile/FOMAML algebra
(variables.py:9-45) becomes axpby on them, and the multi-GPU exchange is one all-reduce over the trainable buffer.
Padding elements are zero and stay zero under every arena operation (SGD: grad 0; axpby: 0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import spec


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class Arena:
    def __init__(self, arch: spec.Arch, device):
        self.arch = arch
        self.device = torch.device(device)
        self.table: List[spec.Param] = spec.param_table(arch)
        self.trainable = [p for p in self.table if p.trainable]
        self.moving = [p for p in self.table if not p.trainable]
        off = 0
        self.t_off: Dict[str, int] = {}
        for p in self.trainable:
            self.t_off[p.name] = off
            off += _pad4(p.size)
        self.n_trainable_padded = off
        self.n_trainable = sum(p.size for p in self.trainable)
        off = 0
        self.m_off: Dict[str, int] = {}
        for p in self.moving:
            self.m_off[p.name] = off
            off += _pad4(p.size)
        self.n_moving_padded = off
        self.theta = torch.zeros(self.n_trainable_padded, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.n_trainable_padded, dtype=torch.float32, device=self.device)
        self.bn_moving = torch.zeros(self.n_moving_padded, dtype=torch.float32, device=self.device)
        mask = np.zeros(self.n_trainable_padded // 4, dtype=np.uint8)
        for p in self.trainable:
            if p.l2:
                o = self.t_off[p.name] // 4
                mask[o:o + _pad4(p.size) // 4] = 1
        self.l2_quad_mask = torch.from_numpy(mask).to(self.device)
        self.by_name = {p.name: p for p in self.table}
        self.w: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.mv: Dict[str, torch.Tensor] = {}
        for p in self.trainable:
            o = self.t_off[p.name]
            self.w[p.name] = self.theta[o:o + p.size].view(p.shape)
            self.g[p.name] = self.grad[o:o + p.size].view(p.shape)
        for p in self.moving:
            o = self.m_off[p.name]
            self.mv[p.name] = self.bn_moving[o:o + p.size].view(p.shape)
        self.reset_moving()

    # ------------------------------------------------------------------ init (efficientnet_model.py:61-82; TF defaults)
    def reset_moving(self):
        self.bn_moving.zero_()
        for p in self.moving:
            if p.kind == "moving_variance":
                self.mv[p.name].fill_(1.0)

    def init_weights(self, seed: int = 0):
        """conv_kernel_initializer N(0, sqrt(2/(k*k*Cout))) for backbone/SE/final convs; glorot-uniform for the RSD convs
        (tf.layers.conv2d default); biases 0; gamma 1, beta 0.  Host numpy RNG (reproducible across devices)."""
        g = np.random.default_rng(seed)
        host = np.zeros(self.n_trainable_padded, dtype=np.float32)
        for p in self.trainable:
            o = self.t_off[p.name]
            if p.init == "normal_fanout":
                v = g.standard_normal(p.shape) * math.sqrt(2.0 / (p.shape[0] * p.shape[1] * p.shape[3]))
            elif p.init == "glorot_uniform":
                rf = p.shape[0] * p.shape[1]
                lim = math.sqrt(6.0 / (rf * p.shape[2] + rf * p.shape[3]))
                v = g.uniform(-lim, lim, p.shape)
            elif p.init == "ones":
                v = np.ones(p.shape)
            else:
                v = np.zeros(p.shape)
            host[o:o + p.size] = np.asarray(v, dtype=np.float32).reshape(-1)
        self.theta.copy_(torch.from_numpy(host))
        self.reset_moving()

    # ------------------------------------------------------------------ packed (unpadded, TF order) import / export
    def _pack(self, buf: torch.Tensor, params, offs) -> torch.Tensor:
        return torch.cat([buf[offs[p.name]:offs[p.name] + p.size] for p in params])

    def _unpack(self, flat: torch.Tensor, buf: torch.Tensor, params, offs):
        flat = flat.to(device=self.device, dtype=torch.float32).reshape(-1)
        if flat.numel() != sum(p.size for p in params):
            raise ValueError("expected {} values, got {}".format(sum(p.size for p in params), flat.numel()))
        o = 0
        for p in params:
            buf[offs[p.name]:offs[p.name] + p.size].copy_(flat[o:o + p.size])
            o += p.size

    def export_trainable_packed(self) -> torch.Tensor:
        return self._pack(self.theta, self.trainable, self.t_off)

    def import_trainable_packed(self, flat: torch.Tensor):
        self._unpack(flat, self.theta, self.trainable, self.t_off)

    def export_grad_packed(self) -> torch.Tensor:
        return self._pack(self.grad, self.trainable, self.t_off)

    def export_moving_packed(self) -> torch.Tensor:
        return self._pack(self.bn_moving, self.moving, self.m_off)

    def import_moving_packed(self, flat: torch.Tensor):
        self._unpack(flat, self.bn_moving, self.moving, self.m_off)

    def named_numpy(self) -> Dict[str, np.ndarray]:
        out = {p.name: self.w[p.name].detach().cpu().numpy().copy() for p in self.trainable}
        out.update({p.name: self.mv[p.name].detach().cpu().numpy().copy() for p in self.moving})
        return out

    def load_named(self, values: Dict[str, np.ndarray], strict: bool = True, prefixes: Optional[List[str]] = None,
                   exclude_prefix: Optional[str] = None) -> int:
        """Restore by variable name with the scope filters of EfficientLab.restore_model (models/efficientlab.py:425-430)."""
        n = 0
        for p in self.table:
            if prefixes is not None and not any(p.name.startswith(x) for x in prefixes):
                continue
            if exclude_prefix is not None and p.name.startswith(exclude_prefix):
                continue
            if p.name not in values:
                if strict:
                    raise KeyError("variable {} missing from checkpoint".format(p.name))
                continue
            v = torch.from_numpy(np.asarray(values[p.name], dtype=np.float32)).reshape(p.shape)
            (self.w if p.trainable else self.mv)[p.name].copy_(v)
            n += 1
        return n
