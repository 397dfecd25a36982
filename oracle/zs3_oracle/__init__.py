"""zs3_oracle -- CPU restatement of the ZS3 training hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker* for the MI355X-native product in ``zs3_amd``: it restates, in
plain fp32 PyTorch ops on the CPU, what valeoai/ZS3 computes on its hot path (DeepLabv3+
forward/backward, GMMN generator, MMD loss, the supervised and GMMN training steps).  It is
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product package must never import it.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against
golden vectors produced by importing the reference itself in the build container
(tools/make_goldens.py -> tests/golden/*.npz).  Each function cites the reference file:line it
follows (paths relative to the reference checkout).
"""
from .nets import DeepLab, ResNet101Dilated, ASPP, Decoder, Bottleneck  # noqa: F401
from .gmmn import GMMNnetwork  # noqa: F401
from .losses import SegmentationLosses, GMMNLoss, cross_entropy_2d, cross_entropy_2d_closed_form, mmd_loss  # noqa: F401
from .gcn import cluster_graph, gcn_forward, GraphConvolution, GMMNnetwork_GCN  # noqa: F401
from .steps import (  # noqa: F401
    supervised_step,
    gmmn_step,
    gcn_context_step,
    poly_lr,
    apply_lr,
    make_synthetic_batch,
    nearest_index,
    confusion_matrix,
    miou_from_confusion,
    pixel_accuracy,
    fw_iou,
)
