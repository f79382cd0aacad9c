"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's multi_pose inference hot path:
DLA-34 (+DCNv2) forward, heat-map decode, post-process and an SMPL/LBS stage.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and only as the checker / reported CPU baseline.  The product package
(`human-3d-reconstruction_amd/`, imported as `h3d_amd`) never imports it and has no CPU
fallback: it raises if the HIP library is missing.

Parity status:
  * dla.py, decode.py     -- pinned by golden vectors produced by importing the reference's own
                             code (oracle/gen_golden.py -> tests/golden/*.npz).
  * dcn.py / dcn_ref.c    -- the reference's DCNv2 has no CPU path and cannot be built here;
                             pinned by the reference's known-answer test (DCNv2/test.py:32-67)
                             and derived identities.
  * post_process.py       -- reference module needs cv2 (absent): pinned by analytic identities.
  * smpl.py               -- NO reference code exists for SMPL: **parity unpinned**; checked by
                             analytic known answers only.
"""
