"""CPU oracle for the SynthMorph/VoxelMorph hot path -- TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``multimodal-registration_amd/`` imports it and the
product path raises when the HIP library is missing.

PARITY UNPINNED for the tensor operators.  The reference
(ivadomed/multimodal-registration) ships no tests, golden vectors or fixtures
 and the arithmetic lives in un-vendored third-party
packages that are absent from this image (voxelmorph @ 52dd120f, neurite @
c7bb05d5, pystrum @ 8cd5c483, TensorFlow 2.7 -- README.md:35-42 of the
reference).  ``ops_np`` therefore restates the *published* algorithm of those
packages as written down in SURVEY.md Appendix A, anchored on the reference's
call sites (train_synthmorph.py:57-67,288-307; 3d_reg.py:305-334,377-394;
bids_two_steps_registration.py:324).  What pins it:

* analytic known-answer tests (tests/test_oracle_kat.py),
* cross-checks against independent torch-CPU implementations of the same
  maths (grid_sample/interpolate/conv3d/max_pool3d),
* for the NumPy host helpers the reference implements itself
  (get_def_field_from_subvol, set_random_zero_borders, gen_synthmorph_eb, the
  tiling arithmetic) golden vectors generated from the reference with
  tests/golden/make_golden.py -- those ARE pinned.
"""
