"""CPU oracle for the patch-refined depth inference path.  TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32 restatement of the reference algorithm (zhyever/PatchRefinerV2),
function by function, each citing the reference file:line it follows.  It exists to
*check* the HIP path; nothing under ``patchrefinerv2_amd/`` imports it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning: ``oracle/make_golden.py`` (run in the build container only, where
``/root/reference`` exists) imports the reference Python under ``sys.modules`` stubs,
feeds reference module and oracle the same synthetic weights and inputs, asserts they
agree, and writes the reference outputs to ``tests/golden/*.npz``.  The CPU test-suite
re-checks the oracle against those committed vectors.

Parity-unpinned pieces (arithmetic lives in third-party packages absent from
/root/reference; restated from their published semantics):
  * torchvision.ops.roi_align 0.16.2      -> oracle.ops.roi_align
  * cv2.GaussianBlur 4.8.1                 -> oracle.ops.gaussian_blur
  * timm mobilenetv4_conv_small            -> oracle.mnv4
  * MiDaS DPT_BEiT_L_384 (torch.hub)       -> not restated
"""
