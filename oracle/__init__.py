"""CPU oracle for the FLUX-Kontext denoiser hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU) restatement of the arithmetic the reference
(wyhlovecpp/GPT-Image-Edit) reaches through the un-vendored third-party package
``diffusers==0.32.2`` (reference ``requirements.txt:22``) plus the pure helpers of the
reference's vendored pipeline driver ``univa/utils/flux_pipeline.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it -- always as the checker, never as the thing measured or shipped.  The product
package ``gpt_image_edit_amd`` never imports it and fails loudly without its HIP library.

Pinning status
--------------
* ``oracle.helpers`` (``calculate_shift``, ``_pack_latents``, ``_unpack_latents``,
  ``_prepare_latent_image_ids``, ``dynamic_resize`` family) is PINNED: checked against golden
  vectors produced by executing the reference's own functions in the build container
  (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
* ``oracle.scheduler.shifted_sigmas`` (the dynamic shift) is PINNED to the reference's own in-tree restatement of
  it, ``apply_flux_schedule_shift`` (``train_denoiser.py:972-986``), through ``tests/golden/train.npz``
  (``tests/test_oracle_golden.py::test_scheduler_shift_pinned_to_the_references_own_restatement``); the Euler step and
  the linspace grid remain restatements.
* ``oracle.mmdit`` / ``oracle.vae`` / the rest of ``oracle.scheduler`` are **parity unpinned** against the
  real third-party code: ``diffusers`` is not vendored in the reference and not installed here
  (no network), and the reference holds no tests or golden vectors for this path (SURVEY.md
  section 4 / 8c).  They restate the published diffusers 0.32.2 algorithm,
  are anchored on the reference's call sites (cited in each docstring), and are pinned to torch's
  own CPU definitions of the building-block ops (golden set G5).
"""
