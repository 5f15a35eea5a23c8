// Pins the sampler oracle on the REFERENCE's own deterministic permutation.  Not runnable in the build image (no JVM /
// Spark); a maintainer with the reference's subgraph_sampler assembly on the classpath runs, from the repo root:
//
//   spark-shell --jars <reference>/scala/subgraph_sampler/target/scala-2.12/subgraph_sampler-assembly-1.0.jar \
//               -i scripts/spark/deterministic_sample.scala
//
// It calls libs.task.SamplingStrategy.hashBasedUniformPermutation (scala/subgraph_sampler/src/main/scala/libs/task/
// SamplingStrategy.scala:16-82) exactly as SGSPureSparkV1Task does — 1-hop over (_dst_node, sorted in-neighbour array)
// with the process-global counter at 1 (:313-388), then 2-hop over (_0_hop, _1_hop, sorted in-neighbour array of _1_hop)
// with the counter at 2 (:390-494), samplingSeed = 42, numNeighborsToSample = 3 — on the bidirectionalised edge list of
// the reference's own sampler fixture (tests/golden/spark_input_edges.csv, 16 nodes) and writes
// tests/golden/spark_deterministic_sample.json.  tests/test_spark_golden.py activates as soon as that file exists and
// holds oracle/gigl_oracle.c (CPU) and gigl_sample_khop (GPU) to it, set for set.
import org.apache.spark.sql.{functions => F}
import libs.task.SamplingStrategy

val f = 3
val seed: Integer = 42
val edges = spark.read.option("header", "true").option("inferSchema", "true")
  .csv("tests/golden/spark_input_edges.csv")
  .select(F.col("_src_node").cast("int"), F.col("_dst_node").cast("int"))
// sampleOnehopSrcNodesUniformly: array_sort(collect_list(_src_node)) GROUP BY _dst_node, permute, slice(1, f)
val oneHopArr = edges.groupBy("_dst_node").agg(F.array_sort(F.collect_list("_src_node")).alias("_1_hop_arr"))
  .select(F.col("_dst_node").alias("_0_hop"), F.col("_1_hop_arr"))
val oneHop = SamplingStrategy.hashBasedUniformPermutation(oneHopArr, "_1_hop_arr", seed)   // counter 1
  .select(F.col("_0_hop"), F.slice(F.col("_shuffled_1_hop_arr"), 1, f).alias("_sampled_1_hop_arr"))
// sampleTwohopSrcNodesUniformly: explode hop 1, join the in-neighbour arrays of _1_hop, permute with K = _0_hop + _1_hop
val exploded = oneHop.select(F.col("_0_hop"), F.explode(F.col("_sampled_1_hop_arr")).alias("_1_hop"))
val nbrArr = edges.groupBy("_dst_node").agg(F.array_sort(F.collect_list("_src_node")).alias("_2_hop_arr"))
val twoHopArr = exploded.join(nbrArr, exploded("_1_hop") === nbrArr("_dst_node"))
  .select(F.col("_0_hop"), F.col("_1_hop"), F.col("_2_hop_arr"))
val twoHop = SamplingStrategy.hashBasedUniformPermutation(twoHopArr, "_2_hop_arr", seed)   // counter 2
  .select(F.col("_0_hop"), F.col("_1_hop"), F.slice(F.col("_shuffled_2_hop_arr"), 1, f).alias("_sampled_2_hop_arr"))
val h1 = oneHop.collect().map(r => (r.getInt(0), r.getSeq[Int](1).toList)).sortBy(_._1)
val h2 = twoHop.collect().map(r => (r.getInt(0), r.getInt(1), r.getSeq[Int](2).toList)).sortBy(t => (t._1, t._2))
val json = "{\"fanout\": 3, \"sampling_seed\": 42, \"hop1\": [" +
  h1.map { case (r, a) => s"""{"root": $r, "sampled": [${a.mkString(", ")}]}""" }.mkString(", ") + "], \"hop2\": [" +
  h2.map { case (r, p, a) => s"""{"root": $r, "parent": $p, "sampled": [${a.mkString(", ")}]}""" }.mkString(", ") + "]}"
new java.io.PrintWriter("tests/golden/spark_deterministic_sample.json") { write(json); close() }
println(s"wrote ${h1.length} hop-1 rows, ${h2.length} hop-2 rows")
System.exit(0)
