"""
Reference-side binding of the HIP networks: what a maintainer of simon-ging/coot-videotext adds to
select ``name: transformer_hip`` per network.  Importing this module needs the reference on ``sys.path`` (it subclasses the
reference's own ``RetrievalModelManager``); nothing else in this package imports it.

Plug points used: the network-type string dispatched at ``coot/model_retrieval.py:80-84`` and the model-manager
class the training script passes in (``train_retrieval.py:8-9,70-79``).  The reference holds a ``models.TransformerConfig``
OBJECT per network (``nntrainer/models/transformer_legacy.py:26-97``, built from the YAML section whose keys it pops), not the
raw dict: ``hip_config_from_reference`` turns that object into this package's ``TransformerConfig``.
"""
from __future__ import annotations

from typing import Any, Dict

from .config import TransformerConfig, TransformerTypesConst
from .nets import TransformerHip

HIP_NAME = TransformerTypesConst.TRANSFORMER_HIP  # "transformer_hip": the new value next to TRANSFORMER_LEGACY (:108-110)


def _enc_section(enc) -> Dict[str, Any]:
    """TransformerEncoderConfig object (transformer_legacy.py:83-97) -> the YAML section it was built from."""
    return dict(hidden_dim=enc.hidden_dim, num_layers=enc.num_layers, dropout=enc.dropout, num_heads=enc.num_heads,
                pointwise_ff_dim=enc.pointwise_ff_dim, activation=enc.activation.name, norm=enc.norm.name)


def reference_config_to_section(ref_cfg) -> Dict[str, Any]:
    """models.TransformerConfig object -> one ``net_*`` section in the reference's YAML schema (the inverse of its __init__)."""
    d = dict(name=ref_cfg.name, output_dim=ref_cfg.output_dim, dropout_input=ref_cfg.dropout_input, norm_input=ref_cfg.norm_input,
             positional_encoding=ref_cfg.positional_encoding, add_local_cls_token=ref_cfg.add_local_cls_token,
             use_input_fc=ref_cfg.use_input_fc, selfatn_config=_enc_section(ref_cfg.selfatn), use_output_fc=ref_cfg.use_output_fc,
             use_context=ref_cfg.use_context, weight_init_type=ref_cfg.weight_init_type, weight_init_std=ref_cfg.weight_init_std,
             linear_out=getattr(ref_cfg, "linear_out", False))
    if ref_cfg.use_input_fc:
        fc = ref_cfg.input_fc_config
        d["input_fc_config"] = dict(output_dim=fc.output_dim, num_layers=fc.num_layers, hidden_dim=fc.hidden_dim,
                                    activation_middle=fc.activation_middle.name, activation_output=fc.activation_output.name,
                                    dropout_middle=fc.dropout_middle, dropout_output=fc.dropout_output,
                                    norm_middle=fc.norm_middle.name, norm_output=fc.norm_output.name, residual=fc.residual)
    if ref_cfg.use_context:
        d["crossatn_config"] = _enc_section(ref_cfg.crossatn)
    pc = ref_cfg.pooler_config
    d["pooler_config"] = dict(name=pc.name, hidden_dim=pc.hidden_dim, num_heads=pc.num_heads, num_layers=pc.num_layers,
                              dropout=pc.dropout, activation=pc.activation.name)
    return d


def hip_config_from_reference(ref_cfg, input_dim: int) -> TransformerConfig:
    """The adapter the dispatch needs: reference config object + input width -> TransformerConfig of this package (which raises
    NotImplementedError for the reference options the HIP path does not implement, as the reference does for unknown types)."""
    return TransformerConfig(reference_config_to_section(ref_cfg), input_dim)


def make_hip_network(ref_cfg, input_dim: int) -> TransformerHip:
    """Drop-in for ``models.TransformerLegacy(current_cfg, input_dims[key])`` (coot/model_retrieval.py:82): same constructor
    arguments, same ``forward(features, mask, lengths, hidden_state) -> (pooled, per_token)``, same state-dict names."""
    return TransformerHip(hip_config_from_reference(ref_cfg, input_dim))


def reference_manager_class():
    """``RetrievalModelManager`` of the reference with the patched dispatch, as a subclass (so the reference file itself stays
    untouched): pass it where ``train_retrieval.py:70`` constructs ``ModelManager(cfg)``.  Everything else — encode_visual /
    encode_text, get_all_params, get/set_model_state with the legacy renames, train / eval switches — is inherited."""
    from coot import model_retrieval as ref_mr            # the reference (needs it on sys.path)
    from coot.configs_retrieval import RetrievalNetworksConst
    from nntrainer import models

    class HipRetrievalModelManager(ref_mr.RetrievalModelManager):
        def __init__(self, cfg):
            models.BaseModelManager.__init__(self, cfg)
            input_dims = {
                RetrievalNetworksConst.NET_VIDEO_LOCAL: cfg.dataset_val.vid_feat_dim,
                RetrievalNetworksConst.NET_VIDEO_GLOBAL: cfg.model_cfgs[RetrievalNetworksConst.NET_VIDEO_LOCAL].output_dim,
                RetrievalNetworksConst.NET_TEXT_LOCAL: cfg.dataset_val.text_feat_dim,
                RetrievalNetworksConst.NET_TEXT_GLOBAL: cfg.model_cfgs[RetrievalNetworksConst.NET_TEXT_LOCAL].output_dim,
            }
            for key in RetrievalNetworksConst.values():
                current_cfg = cfg.model_cfgs[key]
                if current_cfg.name == models.TransformerTypesConst.TRANSFORMER_LEGACY:
                    self.model_dict[key] = models.TransformerLegacy(current_cfg, input_dims[key])
                elif current_cfg.name == HIP_NAME:  # <- the one added branch of coot/model_retrieval.py:80-84
                    self.model_dict[key] = make_hip_network(current_cfg, input_dims[key])
                else:
                    raise NotImplementedError(f"Coot model type {current_cfg.name} undefined")

    return HipRetrievalModelManager
