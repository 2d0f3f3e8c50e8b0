"""``load_model`` - same signature, defaults and error behaviour as the reference's
range/load_model.py:16-51, for the RANGE / RANGE+ models."""
from __future__ import annotations

from argparse import Namespace

from .range import LocationEncoder, ShardedLocationEncoder


def load_model(model_name="RANGE+", pretrained_path=None, device="cuda", **kwargs):
    """Load a RANGE / RANGE+ (or plain SatCLIP) location encoder running on MI355X.

    Args:
        model_name: 'RANGE', 'RANGE+', 'SatCLIP' (the encoder alone, range.py:117-122) or one of
            the training-free encoders 'Direct', 'Cartesian_3D', 'Wrap' (range.py:152-173).
        pretrained_path: SatCLIP checkpoint (e.g. satclip-vit16-l40.ckpt); the reference demands
            it for every model name, also those that never read it.
        device: 'cuda' / 'cuda:N'.
        **kwargs: ``db_path`` (required) - the range_db_*.npz bank; ``beta`` (RANGE+, default 0.5).
            Optional, not in the reference: ``sh_eval`` - 'reference' (default: 'analytic'
            checkpoints evaluate the reference's own generated polynomials, so embeddings agree
            with the reference at every latitude) or 'exact' (the mathematically exact basis);
            ``sh_source`` - a generated ``spherical_harmonics_ylm.py`` to take the polynomial
            coefficients from (default: regenerated, range_amd/sh_table.py); ``pv_mode`` -
            'exact' (default: float32 products like the reference, range.py:217/:236) or the
            opt-in 'bf16x3' (retrieval products on the bf16 matrix cores, both operands split in
            three planes: ~1.5x the throughput, agreement with the exact products to ~1e-7);
            ``shards`` - W (or True): the bank ROW-SHARDED over the W ranks of the initialised
            ``torch.distributed`` job (one process per GPU, RCCL); every rank calls ``load_model``
            with the same arguments and ``device`` = its own GPU and gets a model with the same
            call contract (``range_amd.range.ShardedLocationEncoder``); ``group`` - the process
            group to shard over (default: the world); ``row_shards`` - R < W: the 2-D layout, the
            bank row-sharded over groups of R ranks, W / R groups each serving its own queries
            (``range_amd.dist.make_layout``; default R = W).
    """
    if pretrained_path is None:
        raise ValueError("Please provide the pretrained model path.")      # load_model.py:31-32
    if "RANGE" in model_name:
        assert "db_path" in kwargs, "db_path is required for RANGE model."  # :34
        db_path = kwargs.get("db_path")
        beta = kwargs.get("beta") if "beta" in kwargs else 0.5             # :37-40
    else:
        db_path = None
        beta = None
    args = Namespace(location_model_name=model_name, pretrained_path=pretrained_path,
                     device=device, range_db=db_path, beta=beta)           # :45-46
    for opt in ("sh_eval", "sh_source", "pv_mode", "shards", "row_shards"):
        if opt in kwargs:
            setattr(args, opt, kwargs[opt])
    if kwargs.get("shards"):
        if "RANGE" not in model_name:
            raise ValueError("shards= applies to the RANGE / RANGE+ models (the bank is what is sharded)")
        model = ShardedLocationEncoder(args, group=kwargs.get("group"))
    else:
        model = LocationEncoder(args)
    model.eval()                                                           # :49
    model.to(model.engine.device)   # :50 (the mirror parameters of loc_model already live there: a no-op that keeps the call)
    return model
