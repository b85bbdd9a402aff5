#!/bin/sh
# download_data.sh — RefSeq sequences for one of the standard databases (needs network access; same directory layout
# as CuCLARK's script, download_data.sh:28-121):  ./download_data.sh <database directory> <bacteria|viruses|human>
# Sequences land in <dir>/Bacteria, <dir>/Viruses or <dir>/Human; the list of files is written to <dir>/.<db>.
if [ $# -lt 2 ] || [ -z "$1" ]; then
  echo "Usage: $0 <Directory for the sequences> <Database: bacteria, viruses or human> "
  exit 0
fi
DBDR=$1
DB=$2
NCBI=ftp://ftp.ncbi.nih.gov/genomes
case "$DB" in
  bacteria) SUB=Bacteria; PAT='*.fna' ;;
  viruses) SUB=Viruses; PAT='*.f??' ;;
  human) SUB=Human; PAT='*.fa' ;;
  *) echo "Failed to recognize parameter: $DB. Please choose between: bacteria, viruses, human."; exit 1 ;;
esac
if [ -s "$DBDR/.$DB" ]; then
  echo "$SUB sequences already in $DBDR."
  exit 0
fi
# start from an empty sequence directory and forget the metadata derived from the old one
if [ -d "$DBDR/$SUB" ]; then
  find "$DBDR/$SUB" -mindepth 1 -delete
fi
mkdir -p -m 775 "$DBDR/$SUB"
for meta in "$DBDR/.$DB.fileToAccssnTaxID" "$DBDR/.$DB.fileToTaxIDs"; do
  [ -f "$meta" ] && unlink "$meta"
done
cd "$DBDR/$SUB" || exit 1
echo "Downloading now $SUB genomes:"
case "$DB" in
  bacteria)
    wget $NCBI/archive/old_refseq/Bacteria/all.fna.tar.gz
    echo "Downloading done. Uncompressing files... "
    tar -zxf all.fna.tar.gz && unlink all.fna.tar.gz ;;
  viruses)
    wget ftp://ftp.ncbi.nlm.nih.gov/genomes/Viruses/all.fna.tar.gz
    wget ftp://ftp.ncbi.nlm.nih.gov/genomes/Viruses/all.ffn.tar.gz
    echo "Downloading done. Uncompressing files... "
    tar -zxf all.fna.tar.gz && unlink all.fna.tar.gz
    tar -zxf all.ffn.tar.gz && unlink all.ffn.tar.gz ;;
  human)
    for c in 01 02 03 04 05 06 07 08 09 10 11 12 13 14 15 16 17 18 19 20 21 22 X Y MT Un; do
      wget "$NCBI/H_sapiens/CHR_$c/hs_ref_GRC*chr${c#0}.fa.gz"
    done
    echo "Downloading done. Uncompressing files... "
    gunzip ./*fa.gz ;;
esac
find "$(pwd)" -name "$PAT" > "../.$DB"
cd ..
if [ ! -s ".$DB" ]; then
  echo "Error: Failed to download $DB sequences. "
  exit 1
fi
echo "$SUB sequences downloaded!"
