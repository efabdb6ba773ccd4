"""PretrainInterface (reference: src/pretrain_interface.py:14-138): config, vocabulary, log directory layout,
resume bookkeeping, DataContainer construction and the text logs -- everything file-visible stays drop-in
(SURVEY Appendix D).  Multi-GPU additions (rank / world_size, shard ownership) are no-ops at world_size 1."""
import pickle
from pathlib import Path
from shutil import rmtree

from .io.dataset import DataContainer
from .marcos import *  # noqa: F401,F403
from .monitor import logger
from .monitor.dashboard import Dashboard
from .monitor.stat import RunningAvgDict
from .parallel import TaskSharder


def load_units(config, model_name):
    """id2units = [<s> | <blank>] + spm units + [</s>] (:33-48) -> 367 entries for the shipped unit file."""
    first = SOS_SYMBOL if model_name == 'transformer' else BLANK_SYMBOL
    units = [first]
    with open(config['solver']['spm_mapping']) as fin:
        units += [line.rstrip().split(' ')[0] for line in fin]
    units.append(EOS_SYMBOL)
    return units


class PretrainInterface:
    def __init__(self, config, paras, id2accent):
        self.config, self.paras = config, paras
        self.train_type = 'pretrain'
        sv = config['solver']
        self.is_memmap, self.is_bucket, self.model_name = paras.is_memmap, paras.is_bucket, paras.model_name
        self.eval_ival, self.log_ival, self.save_ival = sv['eval_ival'], sv['log_ival'], sv['save_ival']
        self.half_batch_ilen, self.dev_max_ilen = sv['half_batch_ilen'], sv['dev_max_ilen']
        self.sample_strategy = paras.sample_strategy
        self.best_cer = self.best_wer = INIT_BEST_ER
        self.sharder = TaskSharder.from_env()

        if paras.model_name not in ('transformer', 'blstm'):
            raise ValueError(f"Unknown model name {paras.model_name}")
        self.id2units = load_units(config, paras.model_name)
        self._metric = None                      # sentencepiece model is loaded on first use (evaluation / probe)

        self.accents = [id2accent[a] for a in paras.pretrain_accents]
        self.num_pretrain = paras.num_pretrain
        self.tgt_accent = id2accent[paras.tgt_accent]
        self.max_step = paras.max_step if paras.max_step > 0 else sv['total_steps']
        assert self.num_pretrain == len(self.accents), \
            f"num_pretrain is {self.num_pretrain}, but got {len(self.accents)} in pretrain_accents"
        self.data_dirs = [Path(sv['data_root']).joinpath(a) for a in self.accents]
        self.log_dir = Path(Path.cwd(), LOG_DIR, self.train_type, sv['setting'], paras.algo, paras.pretrain_suffix,
                            self.tgt_accent, str(paras.runs))
        if not paras.resume:
            if self.sharder.rank == 0:
                if self.log_dir.exists():
                    assert paras.overwrite, f"Path exists ({self.log_dir}). Use --overwrite or change suffix"
                    logger.warning('Overwriting existing directory')
                    rmtree(self.log_dir)
                self.log_dir.mkdir(parents=True)
            self.sharder.barrier()
            self.train_info = RunningAvgDict(decay_rate=0.99)
            self.global_step = 1
        else:
            self.resume_model_path = self.log_dir.joinpath('snapshot.latest')
            info_dict_path = self.log_dir.joinpath('info_dict.latest')
            # the reference also demands optimizer.latest here although pretraining never writes one (SURVEY section 5,
            # quirk Q3); we resume from snapshot.latest + meta_state.latest when the latter exists.
            self.global_step = int(Path(self.log_dir, 'global_step').read_text().strip())
            assert self.resume_model_path.exists(), f"{self.resume_model_path} not exists..."
            assert info_dict_path.exists(), f"PreTraining info {info_dict_path} not exists..."
            with open(info_dict_path, 'rb') as fin:
                self.train_info = pickle.load(fin)
        self.dashboard = Dashboard(config, paras, self.log_dir, self.train_type, paras.resume)

    @property
    def metric_observer(self):
        if self._metric is None:
            from .monitor.metric import Metric
            sos = 0 if self.paras.model_name == 'transformer' else len(self.id2units) - 1
            self._metric = Metric(self.config['solver']['spm_model'], self.id2units, sos, len(self.id2units) - 1)
        return self._metric

    def load_data(self):
        self.id2ch = self.id2units
        if self.sample_strategy != 'normal':
            raise NotImplementedError
        sv = self.config['solver']
        self.data_container = DataContainer(
            self.data_dirs, batch_size=sv['batch_size'], dev_batch_size=sv['dev_batch_size'], is_memmap=self.is_memmap,
            is_bucket=self.is_bucket, num_workers=self.paras.njobs, min_ilen=sv['min_ilen'], max_ilen=sv['max_ilen'],
            half_batch_ilen=sv['half_batch_ilen'],
            device=getattr(self.paras, 'hbm_shards_device', None), lazy_upload=self.sharder.world > 1)

    def write_log(self, k, v):
        if self.sharder.rank != 0:
            return
        with open(self.log_dir.joinpath(k), 'a') as fout:
            print(f"{self.global_step} {v}", file=fout)

    def log_msg(self, lr=None):
        if self.global_step % self.log_ival == 0 and self.sharder.rank == 0:
            logger.log_info(self.train_info, prefix='train')
            self.dashboard.log_info('train', self.train_info)
            if lr is not None:
                self.dashboard.log_other('lr', lr)
