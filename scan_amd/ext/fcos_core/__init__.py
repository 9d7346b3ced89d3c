"""The package name the reference imports its compiled operators under: ``from fcos_core import _C`` (reference
fcos_core/layers/nms.py:4, layers/sigmoid_focal_loss.py:6).  With ``scan_amd/ext`` on sys.path this is the module built from
scan_amd/csrc/fcos_core_C.cpp on libscan_hip.so -- what a reference checkout binds instead of its own csrc/ build
(INTEGRATION.md).  It holds nothing else."""
