"""Config plumbing with the reference's behaviour (src/retina_net/config_utils.py:7-96): the yaml
basename must equal ``checkpoint_name`` (ValueError otherwise), ``setup`` injects the derived
header fields and the data split, and creates ``<data_dir>/outputs/<name>/{checkpoints,logs}``."""
import os
import shutil

import yaml


def data_dir():
    """<repo>/data, like src/core/__init__.py:4-17 of the reference; override with BAYESOD_DATA_DIR."""
    return os.environ.get("BAYESOD_DATA_DIR",
                          os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data"))


def load_yaml(path):
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def _check_config_name(config, yaml_path):
    name = os.path.splitext(os.path.basename(yaml_path))[0]
    if config['checkpoint_name'] != name:
        raise ValueError('Config checkpoint_name must match the yaml file name: %r vs %r'
                         % (config['checkpoint_name'], name))


def setup(config, args, make_dirs=True):
    """``args`` needs ``yaml_path`` and ``data_split`` (run_inference.py:267-282)."""
    _check_config_name(config, args.yaml_path)
    out_dir = os.path.join(data_dir(), 'outputs', config['checkpoint_name'])
    config['checkpoint_dir'] = os.path.join(out_dir, 'checkpoints')
    config['checkpoint_path'] = os.path.join(config['checkpoint_dir'], config['checkpoint_name'])
    config['logs_dir'] = os.path.join(out_dir, 'logs')
    dataset_name = config['dataset_config']['dataset']
    num_classes = len(config['dataset_config'][dataset_name]['training_data_config']['categories'])
    ag = config['dataset_config']['anchor_generator']
    config['model_config']['header']['num_classes'] = num_classes
    config['model_config']['header']['anchors_per_location'] = len(ag['scales']) * len(ag['aspect_ratios'])
    config['dataset_config']['num_classes'] = num_classes
    config['dataset_config']['data_split'] = args.data_split
    if make_dirs:
        os.makedirs(config['checkpoint_dir'], exist_ok=True)
        os.makedirs(config['logs_dir'], exist_ok=True)
        dst = os.path.join(out_dir, os.path.basename(args.yaml_path))
        if os.path.abspath(dst) != os.path.abspath(args.yaml_path):
            shutil.copy(args.yaml_path, dst)
    return config
