"""Config object with the reference's on-disk formats (common/configuration.py:18-59):
`config___<timestamp>.txt` (one `key = value` per line, CRLF) and `config.pkl` = the pickled
`__dict__` at protocol 2, so run directories written by either implementation load in both
(py2 pickles are read with encoding='latin1')."""
from __future__ import annotations

import os
import pickle
import re
from time import localtime, strftime


def natural_keys(text):
    """Sort key that orders embedded integers numerically (common/natural_sort.py)."""
    return [int(c) if c.isdigit() else c for c in re.split(r'(\d+)', str(text))]


class Config(object):
    """ Configuration object."""

    def __init__(self, **kwargs):
        for key, value in sorted(kwargs.items()):
            setattr(self, key, value)

    def save_config_to_file(self):
        params = sorted(self.__dict__.keys(), key=natural_keys)
        lines = ['%s = %s' % (k, self.__dict__[k]) for k in params]
        name = 'config___%s.txt' % strftime('%Y-%m-%d_%H-%M-%S', localtime())
        with open(os.path.join(self.log_path, name), 'w', newline='') as f:
            f.write('\r\n'.join(lines))
        with open(os.path.join(self.log_path, 'config.pkl'), 'wb') as f:
            pickle.dump(self.__dict__, f, 2)

    def overwrite_safety_check(self, overwrite):
        """ Exits if log_path exists but `overwrite` is set to `False`."""
        if os.path.exists(self.log_path):
            if not overwrite:
                print('\nINFO: log_path already exists. Set `overwrite` to True? Exiting now.')
                raise SystemExit
            print('\nINFO: log_path already exists. The directory will be overwritten.')
        else:
            print('\nINFO: log_path does not exist. The directory will be created.')
            os.makedirs(self.log_path)


def load_config(config_filepath):
    with open(config_filepath, 'rb') as f:
        c_dict = pickle.load(f, encoding='latin1')
    return Config(**c_dict)
