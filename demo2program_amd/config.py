"""Model / data configuration: the fields the reference appends to its argparse namespace
(trainer.py:312-335) and reads in Model.__init__ (models/model_full.py:30-57)."""
import argparse


# BASELINE.json configs
PRESETS = {
    # 1: plumbing-size Karel
    'karel_tiny': dict(dataset_type='karel', batch_size=4, k=2, max_demo_len=20, max_program_len=50,
                       h=8, w=8, depth=16, dim_program_token=50, action_space=6, per_dim=5),
    # 2: Karel full model, 1 GPU -- the headline configuration
    'karel': dict(dataset_type='karel', batch_size=32, k=10, max_demo_len=20, max_program_len=50,
                  h=8, w=8, depth=16, dim_program_token=50, action_space=6, per_dim=5),
    # 4: ViZDoom full model, 1 GPU
    'vizdoom': dict(dataset_type='vizdoom', batch_size=32, k=10, max_demo_len=20, max_program_len=32,
                    h=80, w=80, depth=3, dim_program_token=42, action_space=12, per_dim=6,
                    perception_type='simple', level='not_simple'),
    # 5 (per rank): ViZDoom k=25, 16 programs per rank
    'vizdoom_k25': dict(dataset_type='vizdoom', batch_size=16, k=25, max_demo_len=20,
                        max_program_len=32, h=80, w=80, depth=3, dim_program_token=42,
                        action_space=12, per_dim=6, perception_type='simple', level='not_simple'),
}


def make_config(preset='karel', **overrides):
    """A namespace with every field Model reads (models/model_full.py:30-57)."""
    base = dict(
        # trainer.py:247-289 flag defaults
        debug=False, prefix='default', model='full', dataset_path='datasets/karel_dataset',
        checkpoint=None, log_step=10, write_summary_step=100, test_sample_step=100, num_k=10,
        learning_rate=0.001, lr_weight_decay=False, scheduled_sampling=False,
        scheduled_sampling_decay_steps=20000, encoder_rnn_type='lstm', num_lstm_cell_units=512,
        demo_aggregation='avgpool',
        # trainer.py:322-335
        dsl_type='prob', env_type='no_wall', vizdoom_pos_keys=[], vizdoom_max_init_pos_len=-1,
        perception_type='', level=None, test_k=5,
    )
    base.update(PRESETS[preset])
    base.update(overrides)
    base['num_k'] = base['k']
    return argparse.Namespace(**base)


def n_conv(config):
    # models/model_full.py:219-229: three convs, two more for vizdoom
    return 5 if config.dataset_type == 'vizdoom' else 3


CONV_CHANNELS = [16, 32, 48, 48, 48]


def conv_shapes(config):
    """[(H, W, Cin, Cout, Ho, Wo)] per conv layer (3x3, stride 2, TF SAME)."""
    out = []
    h, w, c = config.h, config.w, config.depth
    for l in range(n_conv(config)):
        ho, wo = (h + 1) // 2, (w + 1) // 2
        out.append((h, w, c, CONV_CHANNELS[l], ho, wo))
        h, w, c = ho, wo, CONV_CHANNELS[l]
    return out


def feature_dim(config):
    _, _, _, cout, ho, wo = conv_shapes(config)[-1]
    return ho * wo * cout


def dataset_module(dataset_type):
    """The reader module the reference imports per dataset type (trainer.py:294-300)."""
    if dataset_type == 'karel':
        from .karel_env import dataset_karel as dataset
    elif dataset_type == 'vizdoom':
        from .vizdoom_env import dataset_vizdoom as dataset
    else:
        raise ValueError(dataset_type)
    return dataset


def input_ops_module(dataset_type):
    if dataset_type == 'karel':
        from .karel_env import input_ops_karel as ops
    elif dataset_type == 'vizdoom':
        from .vizdoom_env import input_ops_vizdoom as ops
    else:
        raise NotImplementedError('The dataset related code is not implemented.')
    return ops


def has_dataset(path):
    import os
    return os.path.exists(os.path.join(path, 'data_info.json')) or os.path.exists(os.path.join(path, 'data.hdf5'))


def config_from_dataset(config, dataset):
    """Data dimensions from the first example and the dataset's DSL / environment fields
    (trainer.py:306-335, evaler.py:457-488)."""
    data = dataset.get_data(dataset.ids[0])
    program, _, s_h, test_s_h, a_h, _, _, _, _, _, _, per, _ = data[:13]
    config.dim_program_token, config.max_program_len = int(program.shape[0]), int(program.shape[1])
    config.k, config.test_k, config.max_demo_len = int(s_h.shape[0]), int(test_s_h.shape[0]), int(s_h.shape[1])
    config.h, config.w, config.depth = (int(v) for v in s_h.shape[2:5])
    config.action_space, config.per_dim = int(a_h.shape[2]), int(per.shape[2])
    if config.dataset_type == 'karel':
        config.dsl_type, config.env_type = dataset.dsl_type, dataset.env_type
        config.vizdoom_pos_keys, config.vizdoom_max_init_pos_len = [], -1
        config.perception_type, config.level = '', None
    else:
        config.dsl_type = config.env_type = 'vizdoom_default'
        config.vizdoom_pos_keys = dataset.vizdoom_pos_keys
        config.vizdoom_max_init_pos_len = dataset.vizdoom_max_init_pos_len
        config.perception_type, config.level = dataset.perception_type, dataset.level
    return config
