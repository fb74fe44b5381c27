"""CPU oracle for the DEAL-YOLO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A pure-PyTorch (CPU, fp32) *functional* restatement of the reference algorithm for the
path SURVEY.md section 8 scopes: the DEAL-YOLO forward graph, the detection loss, box
decode + soft-NMS and the training-step arithmetic.  Every function cites the reference
file:line it follows (paths relative to /root/reference/ultralytics).

Rules (enforced by tests/test_layout.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package;
  * nothing under experiment-yolo_amd/ imports it -- the product path is the HIP library and
    fails loudly when that library is missing;
  * it never reads /root/reference at run time.

Pinning: the reference ships no tests or fixtures for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
tests/golden/make_golden.py (which imports /root/reference under stubs) and committed as
tests/golden/*.npz.  tests/test_oracle_vs_golden.py checks every restated function against them.
"""

from . import graph, nn, loss, nms, trainer  # noqa: F401
