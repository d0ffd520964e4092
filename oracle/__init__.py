"""CPU oracle for the PhotoVerse denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``photoverse_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker.

It is a plain PyTorch fp32 eager restatement of the arithmetic the reference
runs on its hot path (reference files cited per function).  Most of that
arithmetic lives in third-party packages that are NOT vendored in the
reference tree and are NOT installable here (no network):

    diffusers==0.27.2      UNet2DConditionModel, Attention, AttnProcessor2_0,
                           DPMSolverMultistepScheduler   (requirements.txt:1)
    transformers==4.40.0   CLIPVisionModel / CLIPTextModel  (requirements.txt:2)

PARITY STATUS
  * ``adapters_ref``  - PINNED: checked against the real
    ``/root/reference/models/adapters.py`` (imported in the build container by
    ``oracle/make_golden.py``; vectors committed under ``tests/golden/``).
  * ``clip_ref``      - PINNED (version-skewed): checked against the installed
    ``transformers`` 5.x ``CLIPVisionModel`` / ``CLIPTextModel`` with shared
    random weights; the dict-input / concept-injection behaviour of
    ``models/clip.py`` is pinned on the worked example in its comments
    (``clip.py:21-23``).
  * ``attention``     - PINNED at primitive level against
    ``torch.nn.functional.scaled_dot_product_attention`` (the primitive the
    reference calls, ``attention_processor.py:317,400``).
  * ``unet_ref`` / ``scheduler_ref`` - **PARITY UNPINNED**: restated from the
    public SD-v1.5 / diffusers-0.27.2 definition; the reference has no tests
    or golden vectors and diffusers cannot be installed here.  Module and
    state-dict names follow diffusers so a networked check is a one-liner.
"""
