"""flashmd_amd: MI355X (gfx950) CGSchNet force-evaluation path behind the flashmd operator surface.

  flashmd_amd.kernels   the reference's ``flashmd.kernels`` operator names on hand-written HIP
  flashmd_amd.schnet    host mirror of the reference's SchNet / GradientsOut with its MLCG_USE_* gates
  flashmd_amd.gptq      W16A16 module swap (reference models/gptq.py)
  flashmd_amd.engine    fused whole-path energy+force engine (no [E,F] tensor in HBM)
  flashmd_amd.simulate  Langevin BAOAB halves, replica sharding and parallel-tempering exchange
"""
__version__ = "0.1.0"
