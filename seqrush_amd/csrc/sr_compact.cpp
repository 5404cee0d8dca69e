// placeholder until compaction lands
