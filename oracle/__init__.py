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
* ``oracle.mmdit`` / ``oracle.vae`` / ``oracle.scheduler`` are **parity unpinned** against the
  real third-party code: ``diffusers`` is not vendored in the reference and not installed here
  (no network), and the reference holds no tests or golden vectors for this path (SURVEY.md
  section 4 / 8c).  They restate the published diffusers 0.32.2 algorithm,
  are anchored on the reference's call sites (cited in each docstring), and are pinned to torch's
  own CPU definitions of the building-block ops (golden set G5).
"""
