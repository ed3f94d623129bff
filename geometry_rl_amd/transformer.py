"""Transformer baseline actor of BASELINE config 1 (rigid_insertion_multi_transformer_trpl) -- drop-in for
``geometry_rl/modules/pyg_models/transformer_vanilla.py`` (configs/algorithm/pyg_agent/model/transformer.yaml: hidden 64, 2 layers,
2 heads, dropout 0).

The reference module is stock ``torch.nn.TransformerEncoder`` with no custom arithmetic, and config 1 is the reference's CPU-runnable
plumbing case (64 envs x 32 steps): there is no kernel content to replace, so this is stock torch as well (it runs wherever its
parameters live) with the reference's parameter names -- ``cls_token``, ``embedding``, ``transformer_encoder_layer`` (the template layer
the reference also registers), ``transformer_encoder.layers.*``, ``fc_out.lins.0`` (PyG ``MLP([h, out], norm=None)`` = one Linear
[upstream PyG 2.5.2]) -- so a reference state_dict loads strictly.  What IS shared with the HEPi path is everything around it: the
observation split / node features (``HyperData`` with ``concat_input_vector=True``: one launch), the ``post_fc`` Gaussian head, the
fused TRPL kernel, the DeepSets critic kernels, the flat-buffer Adam and the recorded policy-update step.
Pinned by tests/golden/tier2d_transformer_post_fc.npz (generated from the reference module, tools/make_golden.py tier2d)."""
import torch
import torch.nn as nn


class _PygMLP(nn.Module):
    """PyG ``MLP(channel_list, norm=None)`` for a two-entry channel list: a single Linear stored as ``lins.0``."""

    def __init__(self, channel_list):
        super().__init__()
        if len(channel_list) != 2:
            raise NotImplementedError("transformer_vanilla.py:38 builds MLP([in, out], norm=None): one layer")
        self.lins = nn.ModuleList([nn.Linear(channel_list[0], channel_list[1])])

    def forward(self, x):
        return self.lins[0](x)


class TransformerVanilla(nn.Module):
    def __init__(self, input_dim_node, output_dim, num_layers=2, num_heads=2, hidden_dim=64, dropout=0.1, concat_global=False,
                 device=None, **ignored):
        super().__init__()
        self.input_dim, self.concat_global = input_dim_node, concat_global
        self.output_dim = output_dim
        self.cls_token = nn.Parameter(torch.randn(1, 1, output_dim), requires_grad=True)
        self.embedding = nn.Linear(input_dim_node, hidden_dim)
        self.transformer_encoder_layer = nn.TransformerEncoderLayer(d_model=hidden_dim, nhead=num_heads, dim_feedforward=hidden_dim,
                                                                    dropout=dropout)
        self.transformer_encoder = nn.TransformerEncoder(self.transformer_encoder_layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.fc_out = _PygMLP([hidden_dim * 2 if concat_global else hidden_dim, output_dim])
        if device is not None:
            self.to(device)

    @property
    def device(self):
        return next(self.parameters()).device

    def one_step(self, graph, u, **ignored):
        """``u``: the dense node-feature tensor [B, n_nodes, d] of ``HyperData(concat_input_vector=True)`` -- the concatenation over
        ``graph.node_types`` the reference builds from its per-type dict (transformer_vanilla.py:59-66) -- or that dict itself.
        Returns the actuator tokens [B * G, output_dim] (``hidden`` of the post_fc policy head)."""
        if isinstance(u, dict):
            B = graph.batch_size if hasattr(graph, "batch_size") else len(graph)
            u = torch.cat([u[t].reshape(B, -1, u[t].shape[-1]) for t in graph.node_types], dim=1)
        B = u.shape[0]
        mask = self.output_mask(graph)
        x = self.embedding(u.to(self.device))
        if self.concat_global:   # transformer_vanilla.py:70-86
            x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
            h = self.transformer_encoder(x.permute(1, 0, 2)).permute(1, 0, 2)
            cls_out = h[:, 0]
            h = h[:, slice(mask.start + 1, mask.stop + 1)]
            h = torch.cat([cls_out[:, None, :].expand(-1, h.shape[1], -1), h], dim=-1)
        else:                    # :87-92
            h = self.transformer_encoder(x.permute(1, 0, 2)).permute(1, 0, 2)[:, mask]
        return self.fc_out(h.reshape(-1, h.shape[-1]))

    @staticmethod
    def output_mask(graph) -> slice:
        """Positions of the ``output_mask_key`` nodes inside one sample's node sequence (base_data.py:38-43)."""
        if hasattr(graph, "output_mask") and isinstance(graph.output_mask, slice):
            return graph.output_mask
        start = 0
        for t in graph.node_types:
            n = graph.nodes_per_sample[t]
            if t == graph.output_mask_key:
                return slice(start, start + n)
            start += n
        raise KeyError(graph.output_mask_key)
