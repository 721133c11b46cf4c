"""CPU oracle for the AGILE3D hot path (forward_backbone + forward_mask).

TEST INFRASTRUCTURE ONLY.  Nothing under ``agile3d_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the reported CPU baseline.

Pinning status
--------------
* decoder  (``oracle/decoder.py``): PINNED.  Checked in the build container against
  the reference's own ``Agile3d.forward_mask`` / ``get_pos_encs`` (imported from
  /root/reference with a stub MinkowskiEngine module, see
  ``tests/golden/make_goldens.py``); the resulting input/output vectors are
  committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_decoder.py``.
* backbone (``oracle/backbone.py``): PARITY UNPINNED against MinkowskiEngine itself.
  The arithmetic lives in the third-party package ``MinkowskiEngine`` (un-pinned:
  ``pip install -U git+https://github.com/NVIDIA/MinkowskiEngine``,
  reference ``installation.md:30``; latest tag v0.5.4) whose source is neither in
  /root/reference nor installable offline, and the reference has no tests or golden
  vectors for it.  The restatement follows ME's published generalized sparse
  convolution (Choy et al., CVPR'19, "4D Spatio-Temporal ConvNets", eq. 3) as used at
  the reference's call sites, and is pinned independently against dense
  ``torch.nn.functional.conv3d / conv_transpose3d / batch_norm`` on densified scenes
  (``tests/test_oracle_backbone.py``) plus hand-computable known-answer cases.
"""
