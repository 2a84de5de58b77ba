"""Checkpoint interchange and the evaluation protocol of the reference.

* checkpoints: the dict ``train.py:202-207`` saves -- ``{'epoch', 'loss', 'state_dict', 'optimizer', 'val_acc'}`` through
  ``common/utils.py:82-94`` (``weight.pth.tar``, copied to ``model_best.pth.tar`` when best).  ``state_dict`` keys are the
  reference's (identical here, tests/test_abi_and_host_cpu.py); a checkpoint written from the wrapped model carries a
  leading ``module.``.
* image-level result: majority vote of the patch predictions of an image (``common/metric.py:20-50``).
* ``evaluate``: ``train.py:21-91`` -- logits averaged over ``test_time`` re-samplings of the graphs
  (``dataset.set_val_epoch(i)``), patch accuracy from the averaged logits, image / binary accuracy from the votes of every
  pass.
"""
import collections
import os
import shutil

import numpy as np
import torch


# ------------------------------------------------------------------ checkpoints (common/utils.py:82-94, train.py:202-207)
def save_checkpoint(state, is_best, fpath='checkpoint.pth.tar'):
    d = os.path.dirname(fpath)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(state, fpath)
    if is_best:
        shutil.copy(fpath, os.path.join(d, 'model_best.pth.tar'))


def load_checkpoint(fpath, map_location='cpu'):
    if not os.path.isfile(fpath):
        raise ValueError("=> No checkpoint found at '%s'" % fpath)
    return torch.load(fpath, map_location=map_location, weights_only=False)


def checkpoint_state(model, optimizer, epoch, loss, val_acc):
    """The dict of train.py:202-207 (``model`` may be the DataParallel wrapper: its ``.module`` is saved, as at :204)."""
    net = model.module if hasattr(model, 'module') else model
    return {'epoch': epoch + 1, 'loss': loss, 'state_dict': net.state_dict(), 'optimizer': optimizer.state_dict(),
            'val_acc': val_acc}


def load_reference_state(model, checkpoint, strict=True):
    """Load a reference-trained checkpoint (the dict above, or a bare state_dict; keys optionally prefixed ``module.``)."""
    sd = checkpoint['state_dict'] if isinstance(checkpoint, dict) and 'state_dict' in checkpoint else checkpoint
    sd = collections.OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    net = model.module if hasattr(model, 'module') else model
    return net.load_state_dict(sd, strict=strict)


# ------------------------------------------------------------------ image-level vote (common/metric.py:20-50)
class ImageLevelVote(object):
    """``ground_truth``: names like ``<image>_grade_<1|2|3>`` (the reference's GROUND_TRUTH lists) or a dict image -> class."""

    def __init__(self, ground_truth):
        if isinstance(ground_truth, dict):
            self.imglist = dict(ground_truth)
        else:
            self.imglist = {name.split('_grade')[0]: int(name.split('_')[-1]) - 1 for name in ground_truth}
        self.prediction = collections.defaultdict(list)

    @staticmethod
    def _image_of(patch_name):
        return patch_name.split('/')[-1].split('_grade')[0]

    def patch_result(self, name, label):
        self.prediction[self._image_of(name)].append(int(label))

    def batch_patch_result(self, names, labels):
        for name, label in zip(names, labels):
            self.patch_result(name, label)

    def final_result(self):
        gt, result = [], []
        for key, value in self.prediction.items():
            gt.append(self.imglist[key])
            result.append(int(np.argmax([value.count(0), value.count(1), value.count(2)])))
        gt, result = np.asarray(gt), np.asarray(result)
        acc = float((gt == result).mean()) if len(gt) else 0.0
        binary_acc = float(((gt > 0) == (result > 0)).mean()) if len(gt) else 0.0
        return acc, binary_acc


# ------------------------------------------------------------------ evaluate (train.py:21-91)
def _forward_local(dp, chunk):
    """Forward of a DataParallel wrapper on a list that is ALREADY this rank's chunk."""
    keep, dp.shard_input = dp.shard_input, False
    try:
        return dp(chunk)
    finally:
        dp.shard_input = keep


def evaluate(loader, model, vote, test_time=1, max_num_examples=None, batch_size=None):
    """``loader`` yields lists of Data carrying ``.y`` and ``.patch_idx`` (index into ``loader.dataset.idxlist``)."""
    was_training = model.training
    model.eval()
    pred_n, labels_n = [], []
    with torch.no_grad():
        for rep in range(test_time):
            if hasattr(loader.dataset, 'set_val_epoch'):
                loader.dataset.set_val_epoch(rep)
            preds, labels, pending = [], [], []
            for batch_idx, data in enumerate(loader):
                if hasattr(model, 'local_chunk'):            # parallel.DataParallel: this rank scores its chunk of the list,
                    data = model.local_chunk(data)           # then every rank receives every rank's results (rank order =
                    ypred = _forward_local(model, data) if len(data) else model._idle_step()       # list order)
                    # (shard_input = False: the loader already hands every rank its OWN list -- or every rank the same one; a gather
                    # would then count every patch world-size times, so each rank scores what it was given)
                    if model.world > 1 and getattr(model, 'shard_input', True):
                        mine = ([loader.dataset.idxlist[int(d.patch_idx)] for d in data],
                                [int(v) for d in data for v in d.y.reshape(-1)], ypred.detach().cpu().numpy())
                        parts = [None] * model.world
                        torch.distributed.all_gather_object(parts, mine, group=model.group)
                        names = [n for p in parts for n in p[0]]
                        labels.append(np.asarray([v for p in parts for v in p[1]], dtype=np.int64))
                        allp = np.concatenate([p[2] for p in parts if len(p[0])], 0)
                        vote.batch_patch_result(names, allp.argmax(1))
                        preds.append(allp)
                        seen = sum(len(l) for l in labels)       # examples scored so far, all ranks (the gathered count)
                        if max_num_examples is not None and seen > max_num_examples:
                            break
                        continue
                else:                                        # bare module: collate here
                    from .data import Batch
                    dev = next(model.parameters()).device
                    # (a CUDA model: collate on the device -- one packed copy + one kernel, data.py / csrc/collate.hip -- instead of
                    # concatenating on the host and copying tensor by tensor)
                    ypred = model(Batch.from_data_list(data, device=dev) if dev.type == 'cuda' else Batch.from_data_list(data))
                # No device-to-host transfer inside the loop: a `.cpu()` per batch makes the host wait for the batch it has just queued
                # before it starts collating the next one (round 5: the protocol ran 10x below the forward-only rate).  The logits stay
                # on the device until the pass is over; names and labels are host data already.
                names = [loader.dataset.idxlist[int(d.patch_idx)] for d in data]
                labels.append(torch.cat([d.y.reshape(-1) for d in data]).cpu().numpy())
                pending.append((names, ypred.detach()))
                if max_num_examples is not None and (batch_idx + 1) * (batch_size or len(data)) > max_num_examples:
                    break
            if pending:                                      # ONE transfer per test-time pass, then the votes in loader order
                host = torch.cat([p[1] for p in pending], 0).cpu().numpy()
                o = 0
                for names, yp in pending:
                    part = host[o:o + yp.shape[0]]
                    o += yp.shape[0]
                    vote.batch_patch_result(names, part.argmax(1))
                    preds.append(part)
            pred_n.append(np.concatenate(preds, 0)[..., np.newaxis])
            labels_n.append(np.concatenate(labels, 0)[..., np.newaxis])
    pred = np.argmax(np.mean(np.concatenate(pred_n, -1), -1), 1)
    lab = np.mean(np.hstack(labels_n), -1)
    img_acc, binary_acc = vote.final_result()
    if was_training:
        model.train()
    return {'patch_acc': float((lab == pred).mean()), 'img_acc': img_acc, 'binary_acc': binary_acc}


# ------------------------------------------------------------------ GEXF export of the assignment matrices (common/utils.py:48-79)
def cluster_labels(assign_matrix_list):
    """Per ORIGINAL node its cluster at every pooling level: level 1 = argmax of its row of S1, level k = the level-k cluster
    of its level-(k-1) cluster (the mapping chain of common/utils.py:55-71).  Returns {'assign_1': [n] ints, ...}."""
    out, prev = {}, None
    for level, s in enumerate(assign_matrix_list, 1):
        s = s.detach().cpu().numpy() if torch.is_tensor(s) else np.asarray(s)
        hard = np.argmax(s, axis=1)
        prev = hard if prev is None else hard[prev]
        out['assign_%d' % level] = prev.astype(np.int64)
    return out


def output_to_gexf(coordinate, adj, assign_matrix_list, path):
    """Write one graph with node attributes ``x``, ``y``, ``assign_1..k`` as GEXF (the reference's visualisation export;
    same signature).  ``coordinate`` [n, 2]; ``assign_matrix_list``: per level the soft assignment of ONE graph ([n, C1],
    [C1, C2], ...; take ``model.assign_matrix[k][b, :n_b]``); ``adj``: the dense [n, n] matrix the reference passes, or -- so
    that no n x n tensor is needed -- an ``edge_index`` [2, E] (integer dtype)."""
    import networkx as nx
    coordinate = coordinate.detach().cpu().numpy() if torch.is_tensor(coordinate) else np.asarray(coordinate)
    n = coordinate.shape[0]
    G = nx.Graph()
    G.add_nodes_from(range(n))
    a = adj.detach().cpu() if torch.is_tensor(adj) else torch.as_tensor(np.asarray(adj))
    if not a.dtype.is_floating_point and a.dim() == 2 and a.shape[0] == 2 and a.shape[1] != 2:      # edge list
        ei = a.numpy()
        G.add_edges_from(((int(u), int(v), {'weight': 1.0}) for u, v in zip(ei[0], ei[1])))
    else:
        assert a.shape[0] == a.shape[1] == n, 'the adjacent matrix should have same row and col'
        a = a.numpy()
        rows, cols = np.nonzero(a)
        G.add_edges_from(((int(u), int(v), {'weight': float(a[u, v])}) for u, v in zip(rows, cols)))
    for name, lab in cluster_labels(assign_matrix_list).items():
        assert lab.shape[0] == n, 'assignment matrices must belong to this graph (rows = its nodes)'
        nx.set_node_attributes(G, {i: int(v) for i, v in enumerate(lab)}, name)
    nx.set_node_attributes(G, dict(enumerate(coordinate[:, 0].tolist())), 'x')
    nx.set_node_attributes(G, dict(enumerate(coordinate[:, 1].tolist())), 'y')
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    nx.write_gexf(G, path)
    return G
