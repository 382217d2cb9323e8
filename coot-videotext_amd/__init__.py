"""
coot-videotext_amd: MI355X (gfx950) implementation of the COOT retrieval training hot path
(simon-ging/coot-videotext: coot/model_retrieval.py, coot/loss_fn.py, coot/trainer_retrieval.py
hooks, nntrainer/models/*) as hand-written HIP kernels behind a C ABI (include/coot_hip.h).
"""
from . import lib  # noqa: F401
from . import synthetic  # noqa: F401
from .config import (RetrievalConfig, RetrievalNetworksConst, TransformerConfig, TransformerTypesConst,  # noqa: F401
                     load_named_config, load_yaml_config_file)
from .loss_fn import (ContrastiveLoss, ContrastiveLossConfig, cycle_consistency_loss, sample_cycle_indices,  # noqa: F401
                      total_contrastive_loss)
from .model_retrieval import (RetrievalDataBatchTuple, RetrievalModelManager, RetrievalTextEmbTuple,  # noqa: F401
                              RetrievalVisualEmbTuple, RetrievalPackedBatchTuple, attach_packed_index, packed_index)
from .nets import TransformerHip, pack_by_count  # noqa: F401
from .retrieval import compute_retrieval, compute_retrieval_cosine  # noqa: F401
from .trainer_retrieval import RetrievalTrainer, make_optimizer  # noqa: F401
from . import lr_scheduler  # noqa: F401,E402
from .dataset_retrieval import BatchArena, DeviceLoader, RetrievalDataPointTuple, collate_fn  # noqa: F401,E402
