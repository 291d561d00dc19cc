"""Regenerates tests/golden/oracle_v1.npz: outputs of the CPU oracle on small seeded inputs.

These are NOT reference-generated vectors (pyannote.audio cannot be imported in the build container:
lightning / pyannote.core / torchaudio / asteroid-filterbanks are absent and there is no network, see
DESIGN.md section 5).  They pin the ORACLE itself -- and, through the GPU tests that compare against the
same file, the HIP path -- against silent drift between rounds, torch versions and host CPUs.
Run from the repository root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import kaldi_fbank, seeded_pyannet, seeded_wespeaker  # noqa: E402
from oracle.pipeline import diarize  # noqa: E402
from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker, synth_conversation  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(2, 1, 160000, generator=g)).clamp(-1, 1)
    seg = seeded_pyannet(seed=1234, num_layers=4)
    emb = seeded_wespeaker(seed=4321)
    masks = (torch.rand(2, 589, generator=g) < 0.7).float()
    with torch.inference_mode():
        logp = seg(wav)
        fb = kaldi_fbank(wav[0, :, :48000] * 32768.0)
        e = emb(wav[:, :, :48000], weights=masks)
    conv, _ = synth_conversation(24.0, seed=3)
    out = diarize(calibrated_pyannet(calib_seconds=40.0), calibrated_wespeaker(calib_seconds=12.0), conv,
                  exclude_overlap=True)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "oracle_v1.npz"),
        seg_logp=logp.numpy()[:, ::19],            # every 19th frame of 2 chunks x 7 classes
        fbank=fb.numpy()[::23],                    # every 23rd frame x 80 mel bins
        embeddings=e.numpy(),
        pipeline_count=out.count.reshape(-1)[::5].astype(np.uint8),
        pipeline_hard_clusters=out.hard_clusters.astype(np.int8),
        pipeline_turns=np.array([(s, t) for s, t, _ in out.diarization], dtype=np.float64),
    )
    print("written", os.path.join(ROOT, "tests", "golden", "oracle_v1.npz"))


if __name__ == "__main__":
    main()
