"""one line of a multi-rank bench result: whole-job value, streaming rates per rank, host copy rate, zero-copy streaming.  usage: rank_line.py <bench.json> <ranks>"""
import json
import sys

j = json.load(open(sys.argv[1]))
e = j["end_to_end_gpu_parse"]
z = e["streaming_zero_copy"]
print("ranks %s: value %.0f Mpixel/s | streaming sum %.0f (min rank %.0f, max rank %.0f) %.2f ms/batch rank 0 | host copy %.1f GB/s rank 0, %.1f all ranks, "
      "submit/end/begin %s ms | zero-copy streaming sum %.0f, %.2f ms/batch, submit/end/begin %s ms | cores of rank 0: %d (%s, node %s)" % (
          sys.argv[2], j["value"], e["streaming_value"], e["streaming_value_min_rank"], e["streaming_value_max_rank"], e["streaming_ms_per_batch"],
          e["host_copy_GBs"], e["host_copy_GBs_all_ranks"], e["streaming_submit_end_begin_ms"], z["value"], z["ms_per_batch"], z["submit_end_begin_ms"],
          e["affinity"]["cores_of_rank0"], e["affinity"]["core_choice"], e["affinity"]["numa_node_of_gpu"]))
