"""CPU oracle for the FusionDepth training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``fusiondepth_amd`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may.  It restates, in plain fp32 PyTorch-on-CPU (and C for the
integer-ish LiDAR scatter), the arithmetic of the reference files

    layers.py, networks/{resnet_encoder,depth_decoder,pose_decoder,pose_cnn}.py,
    trainer.py:321-596 (predict_poses / generate_images_pred / compute_losses),
    gen2channel.py:60-117 (get_4beam_2channel)

Pinning: every function here is checked against golden vectors produced by
importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  One piece is pinned second-hand: the ResNet
trunk lives in torchvision 0.9 (not vendored in the reference, not installed
here, so the reference's own trunk cannot be run); ``oracle.networks.ResNetTrunk``
restates the public ResNet-v1.5 definition and is pinned (a) structurally
(state-dict keys / shapes = torchvision's, parameter counts) and (b) numerically
against an independent implementation of the same architecture that IS in the
image, Hugging Face ``transformers.ResNetModel``: with the same weights the five
feature maps and the input gradient agree to 1e-4 / 1e-3 for ResNet-18 and
ResNet-50 in training and eval mode
(``tests/test_oracle_golden.py::test_resnet_trunk_matches_an_independent_resnet``).
The refiner golden was generated with this trunk inside the reference's Refiner
(``tests/golden/make_golden.py``), for the same reason.
"""
