"""TEST INFRASTRUCTURE (oracle/): the reference's checkpoint variable list -- names and shapes of everything under
`Model/` outside the slim CNN -- derived STATICALLY from the reference's own source text (no TensorFlow here, no import
of the reference: its files are parsed with `ast`).

    python oracle/ref_var_names.py [/root/reference] > tests/golden/ref_var_names.json

What is read from the source (every literal is looked up at the call site cited, and the script fails if the source says
something else):
  src/model.py:51-55          variable scopes 'Model' / 'encoder' / 'decoder'
  src/model_base.py:80-91     legacy head: layer_norm_activate(scope='LN_tanh'), ops.linear(scope='im_embed')
  src/model_base.py:147,231   variable scope 'rnn_decoder'
  src/model_base.py:541-549   Dense(name='output_projection'), get_variable(name='embedding_map')
  src/model_base.py:606-632   the three cells (BasicLSTMCell / LayerNormBasicLSTMCell / GRUCell)
  src/model_base.py:651-689   'rnn_initial_state' (project_hidden), 'rnn_init_input' + 'projection' (first_input)
  common/ops.py:217-231       ops.linear: get_variable 'weight' (+ 'bias')
  common/ops_rnn.py:28-40     _layer_norm_tanh -> scope 'LN_tanh'
  common/ops_rnn.py:441-470   Dense 'query_layer' / 'memory_layer' / 'value_layer'
  common/ops_rnn.py:543-562   variable_scope(None, 'multi_add_attention'), get_variable 'attention_v', 'softmax_temperature'
  common/ops_rnn.py:623       variable_scope(None, 'MultiHeadDot')
  common/ops_rnn.py:735       Dense(name='a_layer')
  common/ops_rnn.py:640       class MultiHeadAttentionWrapperV3 (its layer scope is the snake-cased class name)

[TF-1.9] scoping rules applied to them (un-vendored dependency, tensorflow==1.9.0; restated from its published source):
  R1  tf.layers / RNNCell layers open `variable_scope(None, default_name=<layer name>)` at their FIRST call (or first
      add_weight) under whatever scope is current THEN, and keep that scope afterwards; an unnamed layer is named by
      the snake-cased class name (`MultiHeadAttentionWrapperV3` -> `multi_head_attention_wrapper_v3`).
  R2  tf.contrib.seq2seq.dynamic_decode(scope=None) runs the decoder step under `variable_scope(None, 'decoder')`.
  R3  DropoutWrapper adds no scope; BasicLSTMCell's variables are `basic_lstm_cell/{kernel,bias}`;
      LayerNormBasicLSTMCell's `layer_norm_basic_lstm_cell/kernel` and `.../{input,transform,forget,output,state}/
      {gamma,beta}` (no bias with layer_norm=True); GRUCell's `gru_cell/{gates,candidate}/{kernel,bias}`.
  R4  tf.contrib.layers.layer_norm(scope=S) creates `S/beta`, `S/gamma`; tf.layers.Dense creates `kernel` (+ `bias`).
  R5  the AttentionMechanism base class calls memory_layer (and MultiHeadAttV3.__init__ the value layer) at CONSTRUCTION,
      i.e. under 'rnn_decoder'; query_layer, attention_v, LN_tanh, softmax_temperature and a_layer are created at the
      first STEP, i.e. under 'rnn_decoder/decoder/multi_head_attention_wrapper_v3' (R1, R2).
  R6  with rnn_init_method='first_input' the cell is first called under 'rnn_decoder/rnn_init_input' (R1: its variables
      stay there); with 'project_hidden' its first call is the wrapper's step (R5's scope).
"""
import ast
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'


def _tree(rel):
    with open(os.path.join(REF, rel)) as f:
        return ast.parse(f.read())


def _strs(node):
    return [n.value for n in ast.walk(node) if isinstance(n, ast.Constant) and isinstance(n.value, str)]


def _calls(tree, attr):
    """all Call nodes whose function is named / ends with `attr`"""
    out = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Call):
            f = n.func
            name = f.attr if isinstance(f, ast.Attribute) else f.id if isinstance(f, ast.Name) else None
            if name == attr:
                out.append(n)
    return out


def _kw(call, key, pos=None):
    for k in call.keywords:
        if k.arg == key and isinstance(k.value, ast.Constant):
            return k.value.value
    if pos is not None and len(call.args) > pos and isinstance(call.args[pos], ast.Constant):
        return call.args[pos].value
    return None


def _func(tree, name):
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == name:
            return n
    raise SystemExit('reference source has no %s' % name)


def need(cond, what):
    if not cond:
        raise SystemExit('reference source does not match the derivation: ' + what)


def snake(name):      # tensorflow/python/layers/base.py (1.9): _to_snake_case
    s = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    return re.sub('([a-z])([A-Z])', r'\1_\2', s).lower()


def read_reference():
    lit = {}
    model = _tree('src/model.py')
    scopes = [_kw(c, 'name_or_scope', 0) for c in _calls(model, 'variable_scope')]
    need({'Model', 'encoder', 'decoder'} <= set(scopes), 'src/model.py scopes Model / encoder / decoder')
    mb = _tree('src/model_base.py')
    enc = _func(mb, '_encoder')
    need(any(_kw(c, 'scope') == 'LN_tanh' for c in _calls(enc, 'layer_norm_activate')), 'legacy head LN_tanh')
    need(any(_kw(c, 'scope') == 'im_embed' for c in _calls(enc, 'linear')), 'legacy head im_embed')
    lit['im_embed_dim'] = [k.value.value for c in _calls(enc, 'linear') for k in c.keywords if k.arg == 'output_dim'][0]
    need(any(_kw(c, 'name_or_scope', 0) == 'cnn' for c in _calls(enc, 'variable_scope')), "encoder scope 'cnn'")
    for fn in ('_decoder_rnn', '_decoder_rnn_scst'):
        need(any(_kw(c, 'name_or_scope', 0) == 'rnn_decoder' for c in _calls(_func(mb, fn), 'variable_scope')),
             "%s scope 'rnn_decoder'" % fn)
    wp = _func(mb, '_build_word_projections')
    need(any(_kw(c, 'name') == 'output_projection' for c in _calls(wp, 'Dense')), 'output_projection')
    need('embedding_map' in _strs(wp), 'embedding_map')
    cells = _strs(_func(mb, '_get_rnn_cell'))
    need({'LSTM', 'LN_LSTM', 'GRU'} <= set(cells), 'rnn_name values')
    cell_calls = {c.func.attr for c in _calls(_func(mb, '_get_rnn_cell'), 'BasicLSTMCell') + _calls(_func(mb, '_get_rnn_cell'), 'LayerNormBasicLSTMCell') +
                  _calls(_func(mb, '_get_rnn_cell'), 'GRUCell')}
    need(cell_calls == {'BasicLSTMCell', 'LayerNormBasicLSTMCell', 'GRUCell'}, 'cell classes')
    init = _func(mb, '_get_rnn_init')
    need(any(_kw(c, 'name_or_scope', 0) == 'rnn_init_input' for c in _calls(init, 'variable_scope')), 'rnn_init_input')
    lin = [(_kw(c, 'scope'), [k for k in c.keywords if k.arg == 'bias_init'][0].value) for c in _calls(init, 'linear')]
    need({s for s, _ in lin} == {'rnn_initial_state', 'projection'}, 'rnn init linear scopes')
    need(all(isinstance(b, ast.Constant) and b.value is None for _, b in lin), 'rnn init projections have no bias')
    ops = _tree('common/ops.py')
    names = [_kw(c, 'name', 0) for c in _calls(_func(ops, 'linear'), 'get_variable')]
    need(names == ['weight', 'bias'], 'ops.linear variable names')
    rops = _tree('common/ops_rnn.py')
    need('LN_tanh' in _strs(_func(rops, '_layer_norm_tanh')), '_layer_norm_tanh scope')
    v3 = _func(rops, 'MultiHeadAttV3')
    dn = {(_kw(c, 'name'), _kw(c, 'use_bias')) for c in _calls(v3, 'Dense')}
    need(dn == {('query_layer', False), ('memory_layer', False), ('value_layer', False)}, 'MultiHeadAttV3 Dense layers')
    add = _func(rops, 'MultiHeadAddLN')
    need(any(_kw(c, 'default_name', 1) == 'multi_add_attention' and isinstance(c.args[0], ast.Constant) and c.args[0].value is None
             for c in _calls(add, 'variable_scope')), 'multi_add_attention default-name scope')
    gv = [_kw(c, 'name', 0) for c in _calls(add, 'get_variable')]
    need(gv == ['attention_v', 'softmax_temperature'], 'MultiHeadAddLN variables')
    # softmax_temperature is created OUTSIDE the multi_add_attention block (same indentation as the `with`)
    call = _func(add, '__call__')
    withs = [n for n in call.body if isinstance(n, ast.With)]
    need(len(withs) == 1 and 'softmax_temperature' not in _strs(withs[0]), 'softmax_temperature outside the attention scope')
    dot = _func(rops, 'MultiHeadDot')
    need(any(_kw(c, 'default_name', 1) == 'MultiHeadDot' for c in _calls(dot, 'variable_scope')), 'MultiHeadDot scope')
    need(not _calls(dot, 'get_variable'), 'MultiHeadDot has no variables of its own')
    wr = _func(rops, 'MultiHeadAttentionWrapperV3')
    need(any(_kw(c, 'name') == 'a_layer' and _kw(c, 'use_bias') is False for c in _calls(wr, 'Dense')), 'a_layer')
    sup = [c for c in _calls(wr, '__init__')]
    need(sup and not any(k.arg == 'name' for c in sup for k in c.keywords), 'wrapper passes no layer name (snake-cased class name applies)')
    lit['wrapper_scope'] = snake(wr.name)
    return lit


def decoder_vars(lit, D=512, E=256, V=258, C=2048, Cg=2048, fm_projection='tied', method='add_LN', context_layer=False,
                 init_method='first_input', rnn_name='LSTM', legacy=False):
    A = C if (fm_projection is None and not context_layer) else D
    Cv = C if fm_projection is None else D
    if legacy:
        Cg = lit['im_embed_dim']
    dec = 'Model/decoder/rnn_decoder/'
    step = dec + 'decoder/' + lit['wrapper_scope'] + '/'
    out = {}
    cell_scope = (dec + 'rnn_init_input/') if init_method == 'first_input' else step           # R6
    Wd = E + A + D
    if rnn_name == 'LSTM':
        out[cell_scope + 'basic_lstm_cell/kernel'] = [Wd, 4 * D]
        out[cell_scope + 'basic_lstm_cell/bias'] = [4 * D]
    elif rnn_name == 'LN_LSTM':
        out[cell_scope + 'layer_norm_basic_lstm_cell/kernel'] = [Wd, 4 * D]
        for s in ('input', 'transform', 'forget', 'output', 'state'):
            out[cell_scope + 'layer_norm_basic_lstm_cell/%s/gamma' % s] = [D]
            out[cell_scope + 'layer_norm_basic_lstm_cell/%s/beta' % s] = [D]
    else:
        out[cell_scope + 'gru_cell/gates/kernel'] = [Wd, 2 * D]
        out[cell_scope + 'gru_cell/gates/bias'] = [2 * D]
        out[cell_scope + 'gru_cell/candidate/kernel'] = [Wd, D]
        out[cell_scope + 'gru_cell/candidate/bias'] = [D]
    if init_method == 'first_input':
        out[dec + 'rnn_init_input/projection/weight'] = [Cg, E + A]
    else:
        out[dec + 'rnn_initial_state/weight'] = [Cg, D]
    out[dec + 'memory_layer/kernel'] = [C, D]                                               # R5: at construction
    if fm_projection == 'independent':
        out[dec + 'value_layer/kernel'] = [C, D]
    att = step + ('multi_add_attention/' if method == 'add_LN' else 'MultiHeadDot/')         # R5: at the first step
    out[att + 'query_layer/kernel'] = [D, D]
    if method == 'add_LN':
        out[att + 'attention_v'] = [D]
        out[att + 'LN_tanh/gamma'] = [D]
        out[att + 'LN_tanh/beta'] = [D]
        out[step + 'softmax_temperature'] = []
    if context_layer:
        out[step + 'a_layer/kernel'] = [Cv, D]
    out[dec + 'output_projection/kernel'] = [D, V]
    out[dec + 'output_projection/bias'] = [V]
    out[dec + 'embedding_map'] = [V, E]
    if legacy:
        out['Model/encoder/LN_tanh/beta'] = [C_POOL]
        out['Model/encoder/LN_tanh/gamma'] = [C_POOL]
        out['Model/encoder/im_embed/weight'] = [C_POOL, lit['im_embed_dim']]
    return out


C_POOL = 1024         # pooled channels of the legacy runs' backbone (Inception-V1 Mixed_5c; model_base.py:80-91 squeezes `net`)
CONFIGS = {
    # name: kwargs of decoder_vars (geometry as DecoderSpec.from_config derives it from the reference Config)
    'comic256_v1': dict(C=832, Cg=1024),
    'comic256_v3': dict(C=2048, Cg=2048),
    'word_baseline': dict(V=25599, C=2048, Cg=2048, fm_projection=None),
    'ln_lstm': dict(rnn_name='LN_LSTM'),
    'gru': dict(rnn_name='GRU'),
    'legacy_v1': dict(C=832, legacy=True),
    'project_hidden': dict(init_method='project_hidden'),
    'dot_context_independent': dict(method='dot', context_layer=True, fm_projection='independent'),
}


if __name__ == '__main__':
    lit = read_reference()
    json.dump({'literals': lit, 'configs': {k: decoder_vars(lit, **kw) for k, kw in CONFIGS.items()}}, sys.stdout, indent=1,
              sort_keys=True)
    sys.stdout.write('\n')
