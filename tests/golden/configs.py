"""Scalar configs of the reduced-width golden fixtures (shared by make_golden.py and the tests)."""
SMALL_VB = dict(
    n_feats=80, n_tokens=500, embedding_dim=96, hidden_size=128, intermediate_size=512,
    num_attention_heads=2, num_hidden_layers=4, convpos_width=31, convpos_groups=2, convpos_depth=2,
    sigma_min=1e-4)
