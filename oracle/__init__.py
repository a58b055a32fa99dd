"""CPU oracle: a plain-PyTorch fp32 restatement of the reference CoBEVT hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg
may import it, and only as the checker / the timed CPU baseline — never as part of the product path
(cobevt_amd/ must not import it; the HIP modules raise when the extension is missing instead of falling back).

Every function cites the reference file:line it follows (paths relative to the reference repository root,
DerrickXuNu/CoBEVT).  The oracle works on a flat state_dict (the reference's key names) plus the reference's
config dicts, deliberately avoiding einops so it is an independent restatement: partitions are explicit
reshape/permute.

Pinning: the reference has no tests or golden vectors.  The oracle is pinned against outputs of the reference
itself, generated in the build container by tests/golden/make_golden.py (which imports /root/reference with
torchvision / shapely stand-ins) and committed as fixtures under tests/golden/*.npz;
tests/test_oracle_golden.py replays them.  Arithmetic that lives in third-party code absent from the
reference tree (torchvision 0.12 ResNet / Bottleneck) is restated from its public definition in
oracle/resnet.py and is pinned only self-consistently (same stand-in on both sides): "parity unpinned" for
torchvision's own arithmetic.  EfficientNet-B4 (efficientnet-pytorch 0.7.1, nuScenes backbone) is restated in
oracle/efficientnet.py the same way and is equally unpinned.
"""
