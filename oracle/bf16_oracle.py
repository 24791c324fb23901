"""CPU emulation of the engine's bf16 precisions (BSMS_BF16 / BSMS_BF16_NODES)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference (Eydcao/BSMS-GNN) is fp32 only: these precisions are an extension of the build (include/bsms_hip.h:
bsms_precision; BASELINE.json configs[2] and [4] name bf16) and have NO reference parity target.  "Parity unpinned" by
the reference; what this file pins is the DOCUMENTED arithmetic of the extension, stated on top of the pinned fp32 oracle
(`oracle/bsms_oracle.py`, golden-checked against the reference itself), so that the HIP kernels of those precisions are
compared with an independent CPU statement of what they are supposed to compute -- not with another run of themselves.
Only `tests/` may import it.

BSMS_BF16, inside every GMP block (reference arithmetic: src/ops/basic.py:26-98; the roundings act at :90-94):
  (a) the weights of the D x D Linears 1..H of the edge MLP are rounded to bf16 (round to nearest even) once per call;
  (b) the activation ENTERING each of those Linears is rounded to bf16; products accumulate in fp32; bias, ReLU and
      LayerNorm are fp32 (src/ops/basic.py:6-23);
  (c) the first edge Linear (fiber + the two node projections) is fp32;
  (d) the messages (LayerNorm output of the edge MLP) are rounded to bf16 before the aggregation (src/ops/basic.py:94),
      whose sums are fp32.
BSMS_BF16_NODES adds: every Linear of the node MLP multiplies bf16 operands -- the rows [x, aggr] and the hidden
activations rounded as they enter, weights rounded once -- with fp32 accumulation, bias, ReLU, LayerNorm and residual."""
import torch

from . import bsms_oracle as ro


def bf16(t: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bf16 (ties to even), returned as fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


class EmulatedGMP(ro.GMP):
    """ro.GMP (src/ops/basic.py:26-98) with the documented bf16 roundings."""

    node_level = False

    def forward(self, x, g, pos):
        send, recv = g[0], g[1]
        rel = ro._take_nodes(pos, send) - ro._take_nodes(pos, recv)
        fiber = torch.cat([rel, torch.norm(rel, dim=-1, keepdim=True)], -1)
        if x.dim() == 3 and pos.dim() == 2:
            fiber = fiber.unsqueeze(0).repeat(x.shape[0], 1, 1)
        seq = self.mlp_edge.seq
        a = torch.relu(seq[0](torch.cat([fiber, ro._take_nodes(x, send), ro._take_nodes(x, recv)], -1)))   # (c): fp32
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)][1:]
        for k, m in enumerate(lin):
            a = torch.nn.functional.linear(bf16(a), bf16(m.weight), m.bias)                                # (a), (b)
            if k < len(lin) - 1:
                a = torch.relu(a)
        msg = bf16(torch.nn.functional.layer_norm(a, a.shape[-1:]))                                        # (d)
        aggr = ro.scatter_sum(msg, recv, dim=-2, dim_size=x.shape[-2])
        if not self.node_level:
            return self.mlp_node(torch.cat([x, aggr], -1)) + x
        nlin = [m for m in self.mlp_node.seq if isinstance(m, torch.nn.Linear)]
        a = torch.cat([x, aggr], -1)
        for k, m in enumerate(nlin):
            a = torch.nn.functional.linear(bf16(a), bf16(m.weight), m.bias)
            if k < len(nlin) - 1:
                a = torch.relu(a)
        return torch.nn.functional.layer_norm(a, a.shape[-1:]) + x


def emulate(net: torch.nn.Module, node_level: bool = False) -> torch.nn.Module:
    """Switch every GMP block of an oracle network (BSGMP / BSMS_Simulator) to the emulated precision, in place."""
    for _, mod in list(net.named_modules()):
        if type(mod) in (ro.GMP, EmulatedGMP):
            mod.__class__ = EmulatedGMP
            mod.node_level = node_level
    return net


def restore(net: torch.nn.Module) -> torch.nn.Module:
    """Back to the fp32 oracle."""
    for _, mod in list(net.named_modules()):
        if type(mod) is EmulatedGMP:
            mod.__class__ = ro.GMP
    return net
