"""CPU oracle for the streaming voice-conversion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under conan_amd/ may import this package:
only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it,
and only as the checker / the timed CPU baseline -- never as the product path.

What it is: a PyTorch-CPU fp32 restatement of the reference's algorithm
(each function cites the reference file:line it follows).  The reference is
itself pure PyTorch, so the restatement calls the same torch CPU operators the
reference calls (F.conv1d, F.layer_norm, F.multi_head_attention_forward ...).

Pinning status (see DESIGN.md "Oracle"):
  * oracle.hifigan, oracle.conan  -- PINNED against the reference itself,
    imported in the build container (tools/make_goldens.py, golden vectors in
    tests/golden/), and against the reference's own invariants
    (hifigan_causal.py:550-680 causality / prefix consistency).
  * oracle.emformer -- PARITY UNPINNED by any reference artefact: the
    arithmetic lives in torchaudio==2.5.1 (requirements.txt:3), which is
    neither vendored in the reference nor installed here.  It restates the
    published torchaudio algorithm; tests pin it against an independently
    written dense-attention formulation and, opt-in, against torchaudio when
    that public package is installed.
  * oracle.frontend -- PARITY UNPINNED: the mel front-end's arithmetic lives in librosa (absent here, not vendored
    in the reference); it restates librosa's published STFT / Slaney mel algorithm and is pinned against an
    independent formulation and analytic properties (tests/test_oracle_frontend.py).
"""
