"""Accuracy report in the reference's own terms (CompareAlignments.cpp): per read the node SETS of the true and of the predicted
alignment are intersected and weighed by node length; a read counts as a good match when

    common / (common + false negative + false positive) >= 0.7        (CompareAlignments.cpp:86)

with common = bp of the shared nodes, false negative / positive = bp of the true / predicted path's mappings minus common
(:13-44; the sums run over the mappings, so a node visited twice counts twice there, as in the reference).  Reads present on only
one side are bad matches (:78-82, 95-98)."""


def alignment_identity(real_nodes, predicted_nodes, node_sizes):
    """(common bp, false-negative bp, false-positive bp) of one read; node ids are bigraph ids (Aligner.cpp:83-91 halves them)"""
    common = sum(node_sizes[n] for n in set(real_nodes) & set(predicted_nodes))
    fn = sum(node_sizes[n] for n in real_nodes) - common
    fp = sum(node_sizes[n] for n in predicted_nodes) - common
    return common, fn, fp


def identity_percent(t):
    total = t[0] + t[1] + t[2]
    return t[0] / total if total else 0.0


def compare(truth, predicted, node_sizes, threshold=0.7):
    """truth / predicted: dict read name -> list of bigraph node ids.  Returns dict(good, bad, per_read)"""
    good = bad = 0
    per_read = {}
    for name, real in truth.items():
        if name not in predicted:
            bad += 1
            continue
        t = alignment_identity(real, predicted[name], node_sizes)
        per_read[name] = t
        if identity_percent(t) < threshold:
            bad += 1
        else:
            good += 1
    bad += sum(1 for name in predicted if name not in truth)
    return dict(good=good, bad=bad, per_read=per_read)


def predicted_nodes(result):
    """bigraph node ids of a result of graphaligner_amd.binding (mapping node ids are digraph ids: id // 2)"""
    return [m[0] // 2 for m in result["mappings"]]
