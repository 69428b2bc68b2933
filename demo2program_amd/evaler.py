"""Evaluation driver with the reference's surface (evaler.py): `Evaler(config, dataset)`,
`eval_run()`, `run_single_step`, `log_step_message`, `log_final_message`, the same CLI flags.

The model is built with is_train=False (evaler.py:61): every batch norm normalises with the
moving statistics of the checkpoint; per batch the teacher-forced and greedy decoders run on the
GPU and the program metrics (syntax / exact program / execution accuracy, Karel) on the host.
Checkpoints are this build's own .npz files (trainer.Trainer.save_checkpoint) or TensorFlow V2
checkpoint prefixes of the reference (`model-<step>` next to `model-<step>.index`, read by
tf_checkpoint.py without TensorFlow -- format and variable names follow TF-1.3 conventions but no
TensorFlow-written file was available to check them against).  `--pred_program` writes the reference's text listing
(`[id: ..]\\ngt: ..\\npred: ..\\ngreedy: ..`) and, in place of its HDF5 file (h5py is not in this
image), a .json with the same per-program fields; `--result_data` writes the reference's per-program
groups (program, pred_program, pred_program_len, s_h, test_s_h) as '<id>/<field>' entries of an .npz.
"""
import argparse
import glob
import json
import os
import time

import numpy as np
import torch

from . import kernels as K
from .config import config_from_dataset, dataset_module, has_dataset, input_ops_module, make_config


class Evaler(object):

    @staticmethod
    def get_model_class(model_name):
        if model_name in ('full', 'summarizer', 'synthesis_baseline'):     # variants of one graph
            from .models.model_full import Model
            return Model
        if model_name == 'induction_baseline':
            raise NotImplementedError('induction_baseline: not built (SURVEY 8(f) N4)')
        raise ValueError(model_name)

    def __init__(self, config, dataset):
        self.config = config
        self.dataset_split = getattr(config, 'dataset_split', 'test')
        self.train_dir = getattr(config, 'train_dir', '')
        self.output_dir = getattr(config, 'output_dir', None) or self.train_dir or '.'
        self.batch_size = config.batch_size
        self.dataset = dataset
        if hasattr(dataset, 'get_data'):
            # a reference-style Dataset gets the input ops of evaler.py:45-56 (in order, no shuffling)
            _, self.batch = input_ops_module(config.dataset_type).create_input_ops(
                dataset, self.batch_size, is_training=False, shuffle=False, frames_dtype=np.uint8)
        else:
            self.batch = dataset            # anything with .next()
        Model = self.get_model_class(config.model)
        self.model = Model(config, is_train=False)
        self.global_step = 0
        self.checkpoint = getattr(config, 'checkpoint', '') or ''
        if self.checkpoint == '' and self.train_dir:
            # evaler.py:96-99 takes tf.train.latest_checkpoint(train_dir): here the highest step among
            # this build's model-<step>.npz and the reference's model-<step>.index files
            found = []
            for path in glob.glob(os.path.join(self.train_dir, 'model-*')):
                base = os.path.basename(path)
                for suffix in ('.npz', '.index'):
                    if base.endswith(suffix) and base[6:-len(suffix)].isdigit():
                        found.append((int(base[6:-len(suffix)]), suffix == '.npz',
                                      path if suffix == '.npz' else path[:-len('.index')]))
            self.checkpoint = max(found)[2] if found else ''
            if not found:
                # asked to evaluate a training directory that holds nothing: do not silently
                # report the accuracy of a random initialisation
                raise IOError('no model-<step>.npz / model-<step>.index checkpoint under --train_dir %s'
                              % self.train_dir)
        if self.checkpoint == '':
            print('No checkpoint is given. Just random initialization :-)')
            self.checkpoint_name = 'random_init'
        else:
            self.checkpoint_name = os.path.basename(self.checkpoint)
        max_steps = getattr(config, 'max_steps', 0)
        self.summary_file = (self.checkpoint or os.path.join(self.output_dir, 'random_init')) + \
            '_report_testdata{}_num_k{}.txt'.format(max_steps * self.batch_size, getattr(config, 'num_k', config.k))

    def load_checkpoint(self, path):
        """Weights and BN moving statistics of a Trainer.save_checkpoint file, or of a TensorFlow V2
        checkpoint prefix written by the reference (tf_checkpoint.py; unverified offline)."""
        from . import tf_checkpoint
        if tf_checkpoint.is_tf_checkpoint(path):
            self.global_step = tf_checkpoint.import_checkpoint(path, self.model)
            return
        z = np.load(path)
        P = self.model.params
        P.load({n: z['p/' + n] for n in P.shapes})
        for n in self.model.moving:
            self.model.moving[n][0].copy_(torch.from_numpy(z['moving_mean/' + n]))
            self.model.moving[n][1].copy_(torch.from_numpy(z['moving_var/' + n]))
        if 'global_step' in z:
            self.global_step = int(z['global_step'])

    # ------------------------------------------------------------------ one batch
    def run_single_step(self, batch, step=None, is_train=False):
        """-> the reference's 19-tuple (evaler.py:287-296)."""
        _start_time = time.time()
        batch_chunk = batch.next()
        m = self.model
        if getattr(m, '_ablate', None):
            raise RuntimeError('this model carries a timing-only ablation (%s): its results are invalid -- Evaler refuses'
                               % ','.join(sorted(m._ablate)))
        feed = m.get_feed_dict(batch_chunk, is_training=False)
        m.forward(feed)
        loss, acc = m.report(with_greedy=True)
        # report() has synchronised.  A persistent recurrent launch that gave up a hand-off (shared device, a
        # workgroup not resident) leaves a STICKY status word and invalid results -- for this batch and every later
        # one until it is reset: switch to the per-step kernels for the rest of the evaluation and redo the batch
        err = K.lstm_persist_error(reset=True)
        if err:
            import sys
            print('[demo2program_amd] persistent LSTM kernel gave up a hand-off (status 0x%08x): evaluating on the '
                  'per-step kernels from here on, batch redone' % (err & 0xffffffff), file=sys.stderr)
            K.lstm_set_persistent(False)
            self.persist_fallbacks = getattr(self, 'persist_fallbacks', 0) + 1
            m.forward(feed)
            loss, acc = m.report(with_greedy=True)
        hist = dict(m.report_hist)
        have_rows = bool(getattr(m, '_program_rows', None))
        have_exec = have_rows and 'program_num_execution_correct' in m._program_rows   # needs an environment
        B = self.batch_size
        none = [None] * B
        out = (
            self.global_step, loss, acc, hist,
            m.pred_program.cpu().numpy(), m.program_len.cpu().numpy(),
            m.program_is_correct_syntax if have_rows else none,
            m.greedy_pred_program.cpu().numpy(), m.greedy_pred_program_len.cpu().numpy().reshape(B, 1),
            m.greedy_program_is_correct_syntax if have_rows else none,
            m.ground_truth_program.cpu().numpy(), m.program_len.cpu().numpy(),
            None,                                           # model.output: not materialised here
            batch_chunk.get('id', np.arange(B)) if hasattr(batch_chunk, 'get') else np.arange(B),
            m.program_num_execution_correct if have_exec else none,
            m.program_is_correct_execution if have_exec else none,
            m.greedy_num_execution_correct if have_exec else none,
            m.greedy_is_correct_execution if have_exec else none,
            time.time() - _start_time,
        )
        return out

    # ------------------------------------------------------------------ the loop
    def eval_run(self):
        cfg = self.config
        if self.checkpoint:
            self.load_checkpoint(self.checkpoint)
            print('Loaded from checkpoint!')
        max_steps = getattr(cfg, 'max_steps', 0) or 1
        pred_program = getattr(cfg, 'pred_program', False)
        text_file = log_file = None
        records = {}
        if pred_program:
            os.makedirs(self.output_dir, exist_ok=True)
            base_name = os.path.join(self.output_dir, 'out_{}_{}'.format(self.checkpoint_name, self.dataset_split))
            text_file = open('{}.txt'.format(base_name), 'w')
            log_file = open('{}.log'.format(base_name), 'w')
            dsl = self.model.vocab
        # --result_data (evaler.py:130-162): per program the ground truth, the greedy prediction and
        # its demonstrations.  The reference writes HDF5 groups; this build writes the same fields
        # as '<id>/<field>' entries of an .npz (h5py is not in the training image)
        result = {} if getattr(cfg, 'result_data', False) else None
        loss_all, acc_all, hist_all, time_all = [], [], {}, []
        loss_keys = acc_keys = None
        final_msg = ''
        for s in range(max_steps):
            (step, loss, acc, hist, pred, pred_len, pred_syntax, greedy, greedy_len, greedy_syntax, gt, gt_len, _,
             program_id, num_exec, is_exec, g_num_exec, g_is_exec, step_time) = self.run_single_step(self.batch)
            step_msg = ''
            if not getattr(cfg, 'quiet', False):
                step_msg = self.log_step_message(s, loss, acc, hist, step_time)
            if result is not None:
                for i in range(len(program_id)):
                    pid = str(program_id[i])
                    if pid + '/program' in result:
                        print('Duplicates: {}'.format(pid))
                        continue
                    result[pid + '/program'] = gt[i]
                    result[pid + '/pred_program'] = greedy[i]
                    result[pid + '/pred_program_len'] = np.asarray(greedy_len[i][0])
                    if hasattr(self.dataset, 'get_data'):
                        data = self.dataset.get_data(pid)
                        result[pid + '/s_h'], result[pid + '/test_s_h'] = data[2], data[3]
            if pred_program:
                log_file.write('{}\n'.format(step_msg))
                correctness = ['wrong', 'correct']
                for i in range(self.batch_size):
                    p_str = dsl.intseq2str(np.argmax(pred[i, :, :int(pred_len[i, 0])], axis=0))
                    g_str = dsl.intseq2str(np.argmax(greedy[i, :, :int(greedy_len[i, 0])], axis=0))
                    pid = str(program_id[i])
                    if pid not in records:
                        records[pid] = {
                            'program_prediction': p_str, 'program_syntax': correctness[int(pred_syntax[i])],
                            'greedy_prediction': g_str, 'greedy_syntax': correctness[int(greedy_syntax[i])]}
                        if num_exec[i] is not None:              # execution results need an environment
                            records[pid].update({
                                'program_num_execution_correct': int(num_exec[i]),
                                'program_is_correct_execution': [bool(v) for v in is_exec[i]],
                                'greedy_num_execution_correct': int(g_num_exec[i]),
                                'greedy_is_correct_execution': [bool(v) for v in g_is_exec[i]]})
                    text_file.write('[id: {}]\ngt: {}\npred{}: {}\ngreedy{}: {}\n'.format(
                        pid, dsl.intseq2str(np.argmax(gt[i, :, :int(gt_len[i, 0])], axis=0)),
                        '(error)' if pred_syntax[i] == 0 else '', p_str,
                        '(error)' if greedy_syntax[i] == 0 else '', g_str))
            loss_keys, acc_keys = list(loss.keys()), list(acc.keys())
            loss_all.append(np.array([loss[k_] for k_ in loss_keys]))
            acc_all.append(np.array([acc[k_] for k_ in acc_keys]))
            time_all.append(step_time)
            for hk, hv in hist.items():
                hist_all.setdefault(hk, []).append(hv)
        if not getattr(cfg, 'no_loss', False):
            loss_avg = np.average(np.stack(loss_all), axis=0)
            acc_avg = np.average(np.stack(acc_all), axis=0)
            hist_avg = {hk: np.average(np.stack(hv), axis=0) for hk, hv in hist_all.items()}
            final_msg = self.log_final_message(loss_avg, loss_keys, acc_avg, acc_keys, hist_avg, list(hist_avg.keys()),
                                               float(np.sum(time_all)),
                                               write_summary=getattr(cfg, 'write_summary', False),
                                               summary_file=self.summary_file)
            self.final = dict(loss=dict(zip(loss_keys, loss_avg.tolist())), acc=dict(zip(acc_keys, acc_avg.tolist())),
                              hist={hk: hv.tolist() for hk, hv in hist_avg.items()})
        if pred_program:
            log_file.write('{}\n'.format(final_msg))
            log_file.write('Model class: {}\n'.format(cfg.model))
            log_file.write('Checkpoint: {}\n'.format(self.checkpoint))
            log_file.write('Dataset: {}\n'.format(getattr(cfg, 'dataset_path', '')))
            log_file.close()
            text_file.close()
            with open('{}.json'.format(base_name), 'w') as f:
                json.dump(records, f)
        if result is not None:
            path = getattr(cfg, 'result_data_path', 'result.hdf5')
            path = (path[:-5] if path.endswith('.hdf5') else path) + ('' if path.endswith('.npz') else '.npz')
            np.savez_compressed(path, **result)
            print('Wrote %d programs to %s' % (len({n.split('/')[0] for n in result}), path))
        print('Completed Evaluation.')

    # ------------------------------------------------------------------ messages (reference formats)
    def log_step_message(self, step, loss, acc, hist, step_time, is_train=False):
        if step_time == 0:
            step_time = 0.001
        loss_str = ''.join('{}:{loss: .3f} '.format(k_, loss=loss[k_]) for k_ in sorted(loss.keys()))
        acc_str = ''.join('{}:{acc: .3f} '.format(k_, acc=acc[k_]) for k_ in sorted(acc.keys()))
        hist_str = ''
        for k_ in sorted(hist.keys()):
            hist_str += '{}: ['.format(k_) + ''.join('{acc: .3f}, '.format(acc=h) for h in hist[k_]) + '] '
        msg = ('[{split_mode:5s} step {step:5d}] ' + '{loss_str}' + '{acc_str}' + '{hist_str}' +
               '({sec_per_batch:.3f} sec/batch, {instance_per_sec:.3f} instances/sec)').format(
            split_mode=(is_train and 'train' or 'val'), step=step, loss_str=loss_str, acc_str=acc_str,
            hist_str=hist_str, sec_per_batch=step_time, instance_per_sec=self.batch_size / step_time)
        print(msg)
        return msg

    def log_final_message(self, loss, loss_key, acc, acc_key, hist, hist_key, time, write_summary=False,
                          summary_file=None, is_train=False):
        loss_str = ''
        for key, i in sorted(zip(loss_key, range(len(loss_key)))):
            loss_str += '{}:{loss: .3f} '.format(loss_key[i], loss=loss[i])
        acc_str = ''
        for key, i in sorted(zip(acc_key, range(len(acc_key)))):
            acc_str += '{}:{acc: .3f}\n'.format(acc_key[i], acc=acc[i])
        hist_str = ''
        for key in sorted(hist_key):
            hist_str += '{}: ['.format(key) + ''.join('{acc: .3f}, '.format(acc=h) for h in hist[key]) + ']\n'
        msg = ('[Final Avg Report] \n' + '[Loss] {loss_str}\n' + '[Acc]  {acc_str}\n' + '[Hist] {hist_str}\n' +
               '[Time] ({time:.3f} sec)').format(loss_str=loss_str, acc_str=acc_str[:-1], hist_str=hist_str[:-1],
                                                 time=time)
        print(msg)
        print('Model class: %s' % self.config.model)
        print('Checkpoint: %s' % self.checkpoint)
        print('Dataset: %s' % getattr(self.config, 'dataset_path', ''))
        if write_summary:
            final_msg = 'Model class: {}\nCheckpoint: {}\nDataset: %s {}\n{}'.format(
                self.config.model, self.checkpoint, getattr(self.config, 'dataset_path', ''), msg)
            os.makedirs(os.path.dirname(os.path.abspath(summary_file)), exist_ok=True)
            with open(summary_file, 'w') as f:
                f.write(final_msg)
        return msg


def build_arg_parser():
    """The reference's flags and defaults (evaler.py:362-427).  Without an HDF5-derived dataset under
    --dataset_path this build evaluates generated Karel programs (karel_env/generator.py)."""
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--model', type=str, default='full',
                        choices=['synthesis_baseline', 'induction_baseline', 'summarizer', 'full'])
    parser.add_argument('--dataset_type', type=str, default='karel', choices=['karel', 'vizdoom'])
    parser.add_argument('--dataset_path', type=str, default='datasets/karel_dataset')
    parser.add_argument('--dataset_split', type=str, default='test', choices=['train', 'test', 'val'])
    parser.add_argument('--checkpoint', type=str, default='')
    parser.add_argument('--train_dir', type=str, default='')
    parser.add_argument('--output_dir', type=str, default=None)
    parser.add_argument('--max_steps', type=int, default=0)
    parser.add_argument('--num_k', type=int, default=10)
    parser.add_argument('--batch_size', type=int, default=20)
    parser.add_argument('--encoder_rnn_type', default='lstm', choices=['lstm', 'rnn', 'gru'])
    parser.add_argument('--num_lstm_cell_units', type=int, default=512)
    parser.add_argument('--demo_aggregation', type=str, default='avgpool', choices=['concat', 'avgpool', 'maxpool'])
    parser.add_argument('--no_loss', action='store_true', default=False)
    parser.add_argument('--pred_program', action='store_true', default=False)
    parser.add_argument('--result_data', action='store_true', default=False)
    parser.add_argument('--result_data_path', type=str, default='result.hdf5')
    parser.add_argument('--id_list', type=str)
    parser.add_argument('--unseen_test', action='store_true', default=False)    # unused by the reference too (evaler.py:415)
    parser.add_argument('--quiet', action='store_true', default=False)
    parser.add_argument('--no_write_summary', action='store_true', default=False)
    return parser


class GeneratedKarelBatches(object):
    """`.next()` -> a fresh batch of generated Karel programs with executed demonstrations."""

    def __init__(self, config, seed=321):
        self.config, self.seed, self.i = config, seed, 0

    def next(self):
        from .karel_env.generator import sample_batch
        self.i += 1
        return sample_batch(self.config, seed=self.seed + self.i)


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    preset = 'karel' if args.dataset_type == 'karel' else 'vizdoom'
    config = make_config(preset, batch_size=args.batch_size, k=args.num_k, num_k=args.num_k, model=args.model,
                         dataset_path=args.dataset_path, encoder_rnn_type=args.encoder_rnn_type,
                         num_lstm_cell_units=args.num_lstm_cell_units, demo_aggregation=args.demo_aggregation)
    for n in ('dataset_split', 'checkpoint', 'train_dir', 'output_dir', 'max_steps', 'no_loss', 'pred_program', 'quiet',
              'result_data', 'result_data_path'):
        setattr(config, n, getattr(args, n))
    config.write_summary = not args.no_write_summary
    if has_dataset(config.dataset_path):
        # evaler.py:431-488: the split to evaluate, steps = one pass, dimensions from the data
        splits = dataset_module(config.dataset_type).create_default_splits(
            config.dataset_path, is_train=False, num_k=config.num_k)
        if config.dataset_split not in ('train', 'test', 'val'):
            raise ValueError('Unknown dataset split')
        target = splits[('train', 'test', 'val').index(config.dataset_split)]
        if args.id_list:
            with open(args.id_list) as f:
                target._ids = [s.strip() for s in f.readlines() if s.strip()]
        if not config.max_steps > 0:
            config.max_steps = int(len(target.ids) / config.batch_size)
        config_from_dataset(config, target)
    elif args.dataset_type == 'karel':
        print('no dataset under %s: evaluating on generated Karel programs' % config.dataset_path)
        target = GeneratedKarelBatches(config)
        if config.max_steps == 0:
            config.max_steps = 1        # a generated dataset has no natural end
    else:
        raise IOError('no ViZDoom dataset under %s (and none can be generated without the engine)'
                      % config.dataset_path)
    Evaler(config, target).eval_run()


if __name__ == '__main__':
    main()
