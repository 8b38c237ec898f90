"""Inference driver with the reference's entry points (src/infer_fn.py): `id_to_caption`,
`run_inference`, `evaluate_model` and the same output files (`captions___N.json`,
`outputs___N.pkl`, `infer_speed.txt`; infer_fn.py:166-184).  The Java COCO scorers
(METEOR / SPICE / PTB tokenizer) are outside the hot path: `evaluate_model`
runs inference and hands the JSON to an evaluator callable -- `comic_amd.coco_eval.evaluate_captions`
(native BLEU-1..4 / ROUGE-L / CIDEr, pinned against the reference's Python scorers) when infer.py finds the
annotation file."""
from __future__ import annotations

import json
import os
import pickle
import re
import time

from . import inputs, model as mdl
from .ops import id_to_caption, base_n_to_dec as _baseN_arr_to_dec  # noqa: F401  (re-exported)

pjoin = os.path.join
P_COCO = re.compile(r'(?<=_)\d+')
P_CKPT = re.compile(r'\d+')


def run_inference(config, curr_ckpt_path, device='cuda:0'):
    """Main inference function. Builds and executes the model."""
    ckpt_dir, ckpt_file = os.path.split(curr_ckpt_path)
    ckpt_num = P_CKPT.findall(ckpt_file)[0]
    mdl.reset_default_graph()
    inputs_man = inputs.InputManager(config, is_inference=True)
    try:
        return _inference_loop(inputs_man, curr_ckpt_path, ckpt_dir, ckpt_file, ckpt_num, device)
    finally:
        inputs_man.close()


def _inference_loop(inputs_man, curr_ckpt_path, ckpt_dir, ckpt_file, ckpt_num, device):
    inputs_man.enable_device_preprocess(device)
    c = inputs_man.config
    batch_size = c.batch_size_infer
    c.checkpoint_path = curr_ckpt_path
    c.resume_training = False
    m_infer = mdl.CaptionModel(c, mode='infer', batch_ops=inputs_man.batch_infer, reuse=False, name='inference',
                               device=device)
    m_infer.restore_model()
    filenames = inputs_man.filenames_infer
    num_batches = int(c.split_sizes['infer'] / batch_size)
    raw_outputs = dict(captions={}, attention={}, image_ids={}, beam_size=c.infer_beam_size,
                       max_caption_length=c.infer_max_length, checkpoint_path=curr_ckpt_path,
                       checkpoint_number=ckpt_num)
    coco_json = []
    print('INFO: Graph constructed. Starting inference.')
    start_time = time.time()
    captions = []
    # captions alone (no --save_attention_maps): the decode loops of two batches are in flight (CaptionModel.infer_pipelined)
    batches = m_infer.infer_pipelined(want_attention=bool(getattr(c, 'save_attention_maps', False)))
    for step in range(num_batches):
        word_ids, attn_maps = next(batches)
        captions = id_to_caption(word_ids, c)
        for i, f in enumerate(filenames[step * batch_size:(step + 1) * batch_size]):
            image_id = f.replace('.jpg', '')
            if '@' in image_id:
                image_id = os.path.basename(image_id)
            else:
                found = P_COCO.findall(image_id)
                if isinstance(found, list) and len(found) > 0:
                    image_id = int(found[0])
                else:
                    raise ValueError('Expected `image_id` to be list or string, saw `{}`'.format(type(found)))
            raw_outputs['captions'][f] = captions[i]
            raw_outputs['attention'][f] = attn_maps[i] if attn_maps is not None else None
            raw_outputs['image_ids'][f] = image_id
            coco_json.append(dict(image_id=image_id, caption=str(captions[i])))
    print('\nExample captions:\n{}\n'.format('\n'.join(captions[:3])))
    t = time.time() - start_time
    assert len(filenames) == len(list(set(filenames)))
    assert len(filenames) == len(coco_json)
    if c.save_attention_maps:
        with open(pjoin(c.infer_save_path, 'outputs___{}.pkl'.format(ckpt_num)), 'wb') as f:
            pickle.dump(raw_outputs, f, 2)
    with open(pjoin(c.infer_save_path, 'captions___{}.json'.format(ckpt_num)), 'w') as f:
        json.dump(coco_json, f)
    speed_file = pjoin(c.infer_save_path, 'infer_speed.txt')
    if not os.path.isfile(speed_file):
        out = ['Using GPU #: {}'.format(c.gpu), 'Inference batch size: {}'.format(c.batch_size_infer),
               'Inference beam size: {}'.format(c.infer_beam_size), '']
        with open(speed_file, 'a', newline='') as f:
            f.write('\r\n'.join(out))
    with open(speed_file, 'a', newline='') as f:
        f.write('\r\n{}'.format(len(filenames) / t))
    print('\nINFO: Inference completed. Time taken: {:4.2f} mins\n'.format(t / 60))
    return coco_json


def evaluate_model(config, curr_ckpt_path, scores_combined, valid_ppl_dict=None, test_ppl_dict=None,
                   evaluate_captions=None):
    """Runs inference for one checkpoint and, when an evaluator callable is supplied
    (annotation file, caption json) -> dict of metric scores, aggregates them like the
    reference (metric_scores.txt / .csv)."""
    c = config
    ckpt_file = os.path.split(curr_ckpt_path)[1]
    ckpt_num = int(P_CKPT.findall(ckpt_file)[0])
    coco_json = pjoin(c.infer_save_path, 'captions___{}.json'.format(ckpt_num))
    if c.run_inference:
        if not (os.path.isfile(curr_ckpt_path) or os.path.isfile(curr_ckpt_path + '.npz')
                or os.path.isfile(curr_ckpt_path + '.index')):
            print('WARNING: `{}` not found. Checkpoint skipped.'.format(ckpt_file))
            return None
        if os.path.isfile(coco_json):
            print('INFO: Found caption file `{}`. Skipping inference.'.format(os.path.basename(coco_json)))
        else:
            run_inference(config, curr_ckpt_path)
    if not c.get_metric_score or evaluate_captions is None:
        return None
    results = evaluate_captions(c.annotations_file, coco_json)
    scores_combined[ckpt_num] = results
    metrics = [m for m in ['Bleu_1', 'Bleu_2', 'Bleu_3', 'Bleu_4', 'METEOR', 'ROUGE_L', 'CIDEr', 'SPICE'] if m in results]
    with open(pjoin(c.infer_save_path, 'metric_scores.txt'), 'a', newline='') as f:
        f.write('===================================\r\n%s\r\nBeam size: %d\r\n===================================\r\n'
                % (ckpt_file, c.infer_beam_size))
        f.write('\r\n'.join('{}: {:1.3f}'.format(m, results[m]) for m in metrics) + '\r\n\r\n\r\n')
    with open(pjoin(c.infer_save_path, 'metric_scores.csv'), 'a', newline='') as f:
        f.write('%d,%s\r\n' % (ckpt_num, ','.join('{:1.3f}'.format(results[m]) for m in metrics)))
    return scores_combined
