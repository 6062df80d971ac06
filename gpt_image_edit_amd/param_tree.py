"""Parameter-holding module trees with the reference's module / parameter names.

The reference's training script selects what it trains by MODULE name -- ``for name, module in
lvlm_model.named_modules(): if check_param_is_in_components(name, ...): module.requires_grad_(True)``
(``train_denoiser.py:538-543``) -- saves with ``save_pretrained`` (``:493``) and loads diffusers state dicts by
dotted key.  The HIP models compute through libfk, not through ``nn.Linear.forward``, but they must LOOK like
the diffusers modules from the outside: ``named_modules()`` yields ``transformer_blocks.3.attn.to_q``,
``named_parameters()`` / ``state_dict()`` yield ``transformer_blocks.3.attn.to_q.weight`` with no name mangling.

``build_param_tree`` hangs one ``ParamNode`` per dotted-path component under a root module (numeric components
such as the ``0`` of ``attn.to_out.0`` or ``ff.net.2`` become children named "0" / "2", exactly what
``nn.ModuleList`` / ``nn.Sequential`` do) and registers the tensors as ``nn.Parameter`` leaves.
"""
import torch
from torch import nn


# Writes through raw pointers (``fk_adamw_step`` rewrites the bf16 copy of a parameter, ``zero.ShardedAdamW.step`` the flat
# buffer the parameters are views of) bump no torch version counter: ``ops.adamw_step`` counts them here instead, and
# ``param_versions`` carries the count, so every cache keyed on it (fused QKV copies, W^T caches, prepared conditioning)
# notices an optimiser step even when the caller forgets ``FluxBackward.refresh()`` / ``model.repack()``.
_RAW_WRITE_EPOCH = [0]


def note_raw_write():
    _RAW_WRITE_EPOCH[0] += 1


def raw_write_epoch():
    return _RAW_WRITE_EPOCH[0]


class ParamNode(nn.Module):
    """A module that only holds parameters and child nodes (the arithmetic lives in libfk)."""

    def forward(self, *args, **kwargs):   # pragma: no cover - never called
        raise RuntimeError("ParamNode holds parameters for the HIP path; call the owning model instead")


def build_param_tree(root, state, requires_grad=False):
    """Register ``state`` (dict dotted-name -> tensor) under ``root`` as a tree of ``ParamNode`` modules.
    Returns dict dotted-name -> nn.Parameter (the same objects ``root.named_parameters()`` yields)."""
    pmap = {}
    for name, tensor in state.items():
        parts = name.split(".")
        node = root
        for comp in parts[:-1]:
            child = node._modules.get(comp)
            if child is None:
                child = ParamNode()
                node.add_module(comp, child)
            node = child
        prm = nn.Parameter(tensor, requires_grad=requires_grad)
        node.register_parameter(parts[-1], prm)
        pmap[name] = prm
    return pmap


class ParamTreeMixin:
    """``p(name)`` / ``has(name)`` over the tree built by :func:`build_param_tree` (kept in ``self._pmap``)."""

    def p(self, name):
        return self._pmap[name]

    def has(self, name):
        return name in self._pmap

    def param_versions(self, names):
        """(data_ptr, in-place version) of each named parameter: changes whenever an optimiser, ``load_state_dict``,
        ``.data = ...`` or an all-gather into a flat buffer rewrites it -- plus, when there is anything to watch, the
        count of raw-pointer parameter writes (``fk_adamw_step``), which no version counter sees."""
        pm = self._pmap
        v = tuple((pm[n].data_ptr(), pm[n]._version) for n in names)
        return v + (raw_write_epoch(),) if v else v


def numel_of(params):
    return sum(int(p.numel()) for p in params)


def to_bf16_state(state):
    return {k: v.to(torch.bfloat16) for k, v in state.items()}
