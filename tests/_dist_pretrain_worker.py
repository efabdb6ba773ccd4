"""Worker of tests/test_hip_dist_rehearsal.py: one rank of `pretrain.py --algo fomaml` on the config-3 toy workspace (started by
torch.distributed.run with MASR_DIST_BACKEND=gloo, all ranks on the one GPU of the box); dumps the final meta weights."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch

import pretrain
from masr_amd import fo_meta_interface


def main():
    out_dir, suffix = sys.argv[1], sys.argv[2]
    captured = {}
    orig = fo_meta_interface.FOMetaASRInterface.train

    def train(self):
        orig(self)
        torch.cuda.synchronize()
        captured["meta"] = self._original.cpu().clone()
        captured["step"] = self.global_step
        captured["collective"] = bool(self.sharder.collective)
        captured["backend"] = self.sharder.backend
        captured["native"] = bool(getattr(self.sharder, "native", False))
    fo_meta_interface.FOMetaASRInterface.train = train
    pretrain.main(["--config", "cfg3.yaml", "--pretrain_suffix", suffix, "--pretrain_accents", "af", "au", "en", "us", "--num_pretrain", "4",
                   "--tgt_accent", "ca", "--algo", "fomaml", "--meta_k", "1", "--meta_batch_size", "4", "--max_step", "5", "--njobs", "2",
                   "--overwrite"] + sys.argv[3:])
    torch.save(captured, os.path.join(out_dir, f"{suffix}_r{os.environ.get('RANK', '0')}.pt"))


if __name__ == "__main__":
    main()
