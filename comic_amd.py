"""Import alias for the package directory `comic-compact-image-captioning-with-attention_amd/`
(a hyphenated directory name is not importable): `import comic_amd` and
`import comic_amd.<submodule>` resolve into that directory."""
import os as _os

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                     'comic-compact-image-captioning-with-attention_amd')
__path__ = [_dir]
with open(_os.path.join(_dir, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_dir, '__init__.py'), 'exec'))
