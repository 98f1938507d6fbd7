"""CPU oracle for the EAGLE draft/verify hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (PyTorch-eager) restatement of the
reference algorithm for the hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker or the timed
CPU baseline; nothing under ``eagle_amd/`` imports it and the product path never routes
through it.

Pinning status
--------------
* ``oracle.draft_model`` and ``oracle.loss`` are pinned against the
  reference itself: ``tests/golden/make_golden.py`` imports ``/root/reference/eagle/llama2.py``
  in the build container, runs it on seeded inputs and stores its outputs as fixtures under
  ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement bit-for-bit
  against those fixtures.
* ``oracle.tree`` (rows a18-a23: static tree buffers, expansion loop, greedy acceptance,
  compaction) has NO counterpart inside ``/root/reference`` (``init_tree`` refers to the
  undefined names ``mc_sim_7b_63`` / ``generate_tree_buffers``, eagle/llama2.py:453-455).
  It restates the published upstream EAGLE-1 algorithm those names point at.
  **parity unpinned** for that module: its own hand-derived known answers are the only pins.
"""
